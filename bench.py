#!/usr/bin/env python3
"""bench.py — filtered scan of a 1e9-row Int64 column at 10 % selectivity on MI355X (BASELINE.json config 2).

One "step" = the whole hot path over the resident column, re-evaluated from scratch (dfdb_query_reset): predicate
scan `x > 899999` -> selection bitmap + tile counts (K1), exclusive scan of the counts, compaction to ascending
1-based Int64 row indices (K2), count left on the device (and all-reduced over ranks when --gpus > 1).  The column is generated in HBM
(splitmix64) before the timed region; outputs stay in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Multi-GPU: contiguous block-range shards (rank r owns rows [r*rows, (r+1)*rows)), no data-path collective,
one RCCL all-reduce of the 8-byte count per step ("scaling": "weak").  One process per GPU: under torch.distributed.run the
ranks come from the environment; started plainly with --gpus N > 1 (WORLD_SIZE unset) this script spawns its own N ranks
BEFORE anything touches a GPU (fresh child processes, never a re-exec) and fails if fewer than N complete.
`--exchange lib` takes the per-step all-reduce through the library's own RCCL communicator (dfdb_group_create_rank /
dfdb_group_count, include/dfdb.h) instead of torch.distributed's.

Beside `value` (config 2, never anything else) the same JSON line carries, as extra keys:
  configs         BASELINE.json configs 3 / 4 / 5 at their stated per-GPU sizes, at EVERY N (every rank runs them on its own shard; times are the
                  MAX over ranks): "3" (a, b, x: conjunctive predicate + projection [b, x]), "4" (String equality + materialize, flat), "4_dictionary"
                  (the same with K9's 16-bit codes beside the column), "5_shard" (the mixed Int64 + Float64 + FlatStrings scan, count() + sum(x) in ONE
                  exchange through the library's own group path: dfdb_group_* with its RCCL communicator), "5_shard_dictionary", "5_shard_materialize"
                  ([a, x] left sharded on the devices: dfdb_group_materialize_device).  Each: rows, selected, ms_per_step, per-kernel avg_ms,
                  algorithmic_GB, rows_per_s (whole job) and roofline {achieved, peak, frac} in algorithmic bytes per GPU.
  calibrated_config  only with --placement: config 2 after the engine's opt-in placement calibration (ctx option placement_calibrate = 1), measured AFTER `value`;
                  `value` itself is the library-default configuration (no ctx option set)
                  Round 4 adds "3_computed" (config 3 with the computed x * 2 projection), "interp" (expressions outside the scan kernels: the device interpreter and
                  the same program compiled at run time by hipRTC), "unique", "unique_hash_table", "unique_float_key", "unique_string", "groupreduce", "groupreduce_int_key", "groupreduce_50k_groups", "groupreduce_hot_key", "groupreduce_float_key", "groupreduce_dictionary", "nullable_string_eq" (SURVEY 8f rows).
  decode_scan     N = 1: the decode-inclusive figure (K7 over the column's LZ4 blocks, fused with the predicate; `unfused`: K7 then K1; both decode with the
                  sequence-start index the column's first resident decode recorded — ctx option lz4_index — and `without_index` is the unfused step without it)
  cold            N = 1: the non-resident path — a table written to /dev/shm in the reference's format, open_table rows/s, block-streamed count and materialize
                  (file bytes/s over PCIe against the pinned-copy rate measured in the run), and a clustered predicate showing late materialization
                  (the projection column read only for the blocks with survivors: dfdb_stream_read_stats)
  cpu_baseline    N = 1: the oracle on the host cores
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

SEED = 0x9E3779B97F4A7C15
THRESHOLD = 899_999          # x > c over h mod 1e6  ->  10 % selectivity
HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable); the run reads it from dfdb_ctx_device_info
KERNELS = ["scan_cmp", "scan_terms", "str_match", "dict_scan", "interp_predicate", "jit_predicate", "interp_project", "jit_project", "scan_counts", "compact_indices", "compact_captured", "gather",
           "str_gather_sizes", "str_gather_bytes", "str_compact_captured", "dict_expand_sizes", "dict_expand_bytes", "fill_const_strings",
           "reduce", "reduce_partials", "range_stage"]


def seed_of(k):              # column k of a table uses seed * (k + 1)
    return (SEED * (k + 1)) & 0xFFFFFFFFFFFFFFFF


def cpu_baseline(rows: int, repeats: int):
    """The oracle (C restatement of the reference's block_streams path) on ONE host core over a bounded
    sample of the same workload: LZ4 block decode -> mask -> LogicalIndex -> row indices."""
    import numpy as np
    import time
    from oracle import oracle as O
    from dfdb import ir
    x = O.gen_i64(SEED, 0, rows)
    t = O.Table(block_size=65536)
    t.add_column("x", x)
    del x
    v = t.view().add_predicate((ir.col(0) > THRESHOLD).to_ir())
    best = None
    nsel = 0
    for _ in range(repeats):
        nsel, sec, _ = v.bench_scan(rows // 5)           # indices are written, like the GPU job
        best = sec if best is None else min(best, sec)
    st = t.column_stats(0)
    res = dict(value=rows / best, unit="rows/s", cores=1, kind="port",
               sample=f"{rows} rows ({st['blocks']} LZ4 blocks of 65536, ratio {st['uncompressed'] / st['compressed']:.2f}), "
                      f"best of {repeats}, {nsel} selected; LZ4 decode -> mask -> LogicalIndex -> 1-based Int64 row indices written")
    # context only (SURVEY.md §8d "CPU_MT"): the same scan with the blocks split over every host core, one oracle table per
    # thread (the C calls release the GIL).  The reference itself is single-threaded, so `value` above stays the 1-core figure.
    try:
        from concurrent.futures import ThreadPoolExecutor
        ncpu = os.cpu_count() or 1
        per = (rows // ncpu // 65536) * 65536
        if ncpu > 1 and per > 0:
            views = []
            for k in range(ncpu):
                tk = O.Table(block_size=65536)
                tk.add_column("x", O.gen_i64(SEED, k * per, per))
                views.append((tk, tk.view().add_predicate((ir.col(0) > THRESHOLD).to_ir())))
            import time
            best_mt = None
            with ThreadPoolExecutor(ncpu) as ex:
                for _ in range(2):
                    t0 = time.perf_counter()
                    list(ex.map(lambda tv: tv[1].bench_scan(per // 5)[0], views))
                    dt = time.perf_counter() - t0
                    best_mt = dt if best_mt is None else min(best_mt, dt)
            res["all_cores"] = dict(value=per * ncpu / best_mt, unit="rows/s", cores=ncpu, sample=f"{per * ncpu} rows split over {ncpu} threads")
    except Exception as e:      # context figure only: never fail the bench for it
        res["all_cores"] = dict(error=str(e))
    # the other BASELINE configs on the same one core, bounded samples of the same seeded columns (the reference's path: LZ4 block decode of the required
    # columns, mask, selection stages, projection gather block by block; materialize(::DFView) evaluates the selection twice — count pre-pass, quirk Q8 —
    # and count() / sum() are two iterations of the view)
    try:
        n = max(1_000_000, rows // 4)
        a, b, x = O.gen_i64(seed_of(0), 0, n), O.gen_i64(seed_of(1), 0, n), O.gen_f64(seed_of(2), 0, n)
        t3 = O.Table(block_size=65536)
        t3.add_column("a", a); t3.add_column("b", b); t3.add_column("x", x)
        v3 = t3.view().add_predicate(((ir.col(0) > 683_771) & (ir.col(2) < 632.456)).to_ir()).set_projection([("b", ir.col(1).to_ir()), ("x", ir.col(2).to_ir())])
        t0 = time.perf_counter(); m3 = v3.materialize(); s3 = time.perf_counter() - t0
        t4 = O.Table(block_size=65536)
        t4.add_column("s", O.gen_str(seed_of(0), 0, n)); t4.add_column("a", O.gen_i64(seed_of(1), 0, n))
        v4 = t4.view().add_predicate((ir.col(0) == "sony").to_ir())
        t0 = time.perf_counter(); m4 = v4.materialize(); s4 = time.perf_counter() - t0
        t5 = O.Table(block_size=65536)
        t5.add_column("a", a); t5.add_column("x", O.gen_f64(seed_of(1), 0, n)); t5.add_column("s", O.gen_str(seed_of(2), 0, n))
        v5 = t5.view().add_predicate(((ir.col(0) > 683_771) & (ir.col(1) < 632.456) & (ir.col(2) != "sony")).to_ir()).set_projection([("x", ir.col(1).to_ir())])
        t0 = time.perf_counter(); c5 = v5.nrow(); sx = v5.sum_f64(0); s5 = time.perf_counter() - t0
        res["configs"] = {"sample_rows": n, "cores": 1, "kind": "port",
                          "3": {"rows_per_s": n / s3, "selected": int(len(m3[0]))}, "4": {"rows_per_s": n / s4, "selected": int(len(m4[1]))},
                          "5_shard": {"rows_per_s": n / s5, "count": int(c5), "sum_x": float(sx), "what": "nrow(v) + sum(v.x): two iterations, like the reference"}}
    except Exception as e:
        res["configs"] = dict(error=f"{type(e).__name__}: {e}")
    return res


def launch_ranks(n: int) -> int:
    """WORLD_SIZE unset and --gpus n > 1: be the launcher.  n fresh child processes of this script, one rank each, with the
    environment torch.distributed.run would give them; this parent never imports torch or touches a GPU.  Rank 0's stdout (the
    JSON line) passes through.  Non-zero exit if any rank fails; the others are then stopped by PID.  The rendezvous port is found by
    bind-and-close, which another process can win before rank 0 binds it: a launch that dies within 60 s is retried on a fresh port."""
    import socket
    import subprocess
    import tempfile
    rc = 1
    for attempt in range(3):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        t_start = time.time()
        errf = tempfile.TemporaryFile()                   # rank 0's stderr: forwarded below, and read for the one failure that is worth a retry
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL,
                                          stderr=errf if r == 0 else None))
        rc = 0
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for k in pending:
                        procs[k].terminate()
            time.sleep(0.05)
        errf.seek(0)
        err0 = errf.read().decode(errors="replace")
        errf.close()
        sys.stderr.write(err0)
        if rc == 0 or not (("EADDRINUSE" in err0 or "address already in use" in err0.lower()) and time.time() - t_start < 60):
            return rc
        print(f"bench.py: port {port} was taken before rank 0 bound it; retrying ({attempt + 1}/3)", file=sys.stderr)
    return rc


def decode_scan_leg(dfdb, ctx, t, rows, steps, out_ptr, cap, cnt_ptr, sync):
    """The like-for-like GPU figure for `cpu_baseline` (which includes the LZ4 decode): the column lives in HBM as the reference's
    LZ4 blocks (written by the device encoder, read back through the ordinary file loader with option keep_compressed), and every
    step decodes all of them (K7) before the same scan + compaction.  Extra keys only: never part of `value`."""
    import shutil
    import tempfile
    need = rows * 6          # the column file is ~4.6 B/row
    base = next((d for d in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(d) and shutil.disk_usage(d).free > need + (4 << 30)), None)
    if base is None:
        return {"skipped": "no scratch directory with %d free bytes for the column file" % need}
    d = tempfile.mkdtemp(prefix="dfdb_bench_", dir=base)
    try:
        st = t.save(os.path.join(d, "tb"))
        ctx.set_option("keep_compressed", 1)
        t2 = dfdb.open_table(os.path.join(d, "tb"), ctx=ctx, load=False)
        t2.load()
        ctx.set_option("keep_compressed", 0)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    q2 = t2[("x", lambda x: x > THRESHOLD), dfdb.ALL]._query()
    nsel2 = q2.count()

    def step_unfused():                                   # K7, then the ordinary K1 over the decoded column
        t2.decode_resident("x")
        q2.reset()
        q2.indices_device(out_ptr, cap)
        q2.count_device(cnt_ptr)

    def step_fused():                                     # K7 with the predicate fused in (ctx option decode_on_scan): decode and filter in one pass
        q2.reset()
        q2.indices_device(out_ptr, cap)
        q2.count_device(cnt_ptr)

    def timed(step):
        step()
        ctx.profile(True)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        el = (time.perf_counter() - t0) / steps
        prof = {k: ctx.profile_get(k) for k in ("lz4_decode", "lz4_decode_scan", "scan_cmp")}
        ctx.profile(False)
        return el, {k: (ms / n if n else None) for k, (n, ms) in prof.items()}

    ctx.set_option("lz4_index", 0)                        # the decoder as a first decode of these blocks runs it: no sequence-start index
    el_p, k_p = timed(step_unfused)
    ctx.set_option("lz4_index", 1)                        # (the library's default: the untimed step inside timed() records the index, the timed ones decode with it)
    el_u, k_u = timed(step_unfused)
    ctx.set_option("decode_on_scan", 1)
    try:
        el_f, k_f = timed(step_fused)
        nsel3 = q2.count()
    finally:
        ctx.set_option("decode_on_scan", 0)
    ms7 = k_u["lz4_decode"]
    res = {"rows_per_s": rows / el_f, "ms_per_step": el_f * 1e3, "steps": steps, "selected": nsel2 if nsel3 == nsel2 else [nsel2, nsel3],
           "blocks": -(-rows // 65536), "compressed_bytes": st["compressed"], "ratio": st["uncompressed"] / max(st["compressed"], 1),
           "lz4_decode_scan_avg_ms": k_f["lz4_decode_scan"], "decoded_GBps": rows * 8 / (k_f["lz4_decode_scan"] * 1e-3) / 1e9 if k_f["lz4_decode_scan"] else None,
           "unfused": {"rows_per_s": rows / el_u, "ms_per_step": el_u * 1e3, "lz4_decode_avg_ms": ms7, "scan_cmp_avg_ms": k_u["scan_cmp"],
                       "decoded_GBps": rows * 8 / (ms7 * 1e-3) / 1e9 if ms7 else None},
           "without_index": {"rows_per_s": rows / el_p, "ms_per_step": el_p * 1e3, "lz4_decode_avg_ms": k_p["lz4_decode"],
                             "decoded_GBps": rows * 8 / (k_p["lz4_decode"] * 1e-3) / 1e9 if k_p["lz4_decode"] else None,
                             "what": "the unfused step with ctx option lz4_index = 0: K7 as a first decode of these blocks runs it (candidate decode + chain walk for every superbatch)"},
           "sequence_index_bytes": st["compressed"] // 8,
           "what": "compressed-resident column (reference LZ4 blocks in HBM) -> K7 decode of every block FUSED with the predicate (bitmap + tile counts leave the "
                   "decoder) -> count scan -> K2 indices, per step; `unfused` = K7, then K1 over the decoded column.  The column's first resident decode recorded "
                   "where its LZ4 sequences start (one bit per compressed byte, ctx option lz4_index); these steps decode with that index, `without_index` without"}
    res["blocks_that_failed_to_decode"] = t2.decode_status("x")      # dfdb_table_decode_status: 0, or the figures above are not a decode
    res["resident_GB"] = {k: v / 1e9 for k, v in t2.resident_bytes("x").items()}
    t2.close()
    # ---- the same column COMPRESSED-ONLY (ctx option keep_compressed = 2; SURVEY.md section 8f-2 "without writing decoded blocks to HBM"): LZ4 blocks + the
    # sequence-start index are all the column holds; the step decodes every block into the waves' 64-KB history rings with the predicate applied on the way
    # (bitmap + tile counts the only output), then count scan + K2.  `materialize`: the same selection + [x] gathered out of the blocks that kept a row.
    try:
        d = tempfile.mkdtemp(prefix="dfdb_bench_", dir=base)
        try:
            t.save(os.path.join(d, "tb"))
            ctx.set_option("keep_compressed", 2)
            t3 = dfdb.open_table(os.path.join(d, "tb"), ctx=ctx, load=False)
            t0 = time.perf_counter()
            t3.load()
            sync()
            load_s = time.perf_counter() - t0
        finally:
            ctx.set_option("keep_compressed", 0)
            shutil.rmtree(d, ignore_errors=True)
        q3 = t3[("x", lambda x: x > THRESHOLD), dfdb.ALL]._query()
        nsel4 = q3.count()

        def step_arena():
            q3.reset()
            q3.indices_device(out_ptr, cap)
            q3.count_device(cnt_ptr)
        step_arena()
        ctx.profile(True)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_arena()
        sync()
        el_a = (time.perf_counter() - t0) / steps
        nh, msh = ctx.profile_get("lz4_decode_scan_hist")
        ctx.profile(False)
        rb = t3.resident_bytes("x")
        ms_h = msh / nh if nh else None
        ar = {"rows_per_s": rows / el_a, "ms_per_step": el_a * 1e3, "selected": nsel4, "count_ok": nsel4 == nsel2, "lz4_decode_scan_hist_avg_ms": ms_h,
              "decoded_GBps": rows * 8 / (ms_h * 1e-3) / 1e9 if ms_h else None, "resident_GB": (rb["decoded"] + rb["compressed"]) / 1e9,
              "resident_decoded_GB": rb["decoded"] / 1e9, "load_seconds": load_s,
              "history_rings_MB": ctx.device_info()["compute_units"] * 24 * 65600 / 1e6,
              "what": "compressed-only column (keep_compressed = 2): K7 decodes every block into per-wave 64-KB history rings with the predicate applied on the way "
                      "(no decoded column exists) -> count scan -> K2 indices, per step"}
        import torch
        ox = torch.empty(max(nsel4, 1), dtype=torch.int64, device="cuda")
        from dfdb import _native as N
        outs = (N.OutCol * 1)()
        outs[0].data, outs[0].memkind = ox.data_ptr(), N.MEM_DEVICE
        lib = N.load()

        def step_mat():
            q3.reset()
            q3.execute()
            N.check(lib.dfdb_materialize(q3._h, outs, 1))
        step_mat()
        sync()
        t0 = time.perf_counter()
        for _ in range(max(2, steps // 2)):
            step_mat()
        sync()
        ar["materialize_x_ms_per_step"] = (time.perf_counter() - t0) / max(2, steps // 2) * 1e3
        res["arena"] = ar
        del ox
        t3.close()
    except Exception as e:
        res["arena"] = {"error": f"{type(e).__name__}: {e}"}
    return res



def cold_leg(dfdb, ctx, torch, dev, rows, peak, chunk_blocks=1024):
    """The NON-resident path (every real DataFrameDBs table starts on disk): a three-column table written in the reference's block format to /dev/shm
    (device LZ4 encoder), then — nothing resident — (1) open_table + load of one column, (2) block-streamed count, (3) block-streamed materialize with an
    unclustered predicate (every block keeps rows: the projection column is read whole), (4) the same with a CLUSTERED predicate (i > 0.9 n over the
    row-number column): late materialization (blocksiterator.jl:111-113) reads the projection column only for the blocks with survivors.  File bytes per
    second are priced against the pinned host-to-device copy rate measured here.  Extra keys only: never part of `value`."""
    import ctypes as C
    import shutil
    import tempfile
    from dfdb import _native as N
    need = rows * 16
    base = next((d for d in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(d) and shutil.disk_usage(d).free > need + (4 << 30)), None)
    if base is None:
        return {"skipped": "no scratch directory with %d free bytes for the table files" % need}
    lib = N.load()
    res = {"rows": rows, "chunk_blocks": chunk_blocks, "files_in": base}
    # the pinned-copy ceiling of this box: 1 GiB pinned -> device, best of 5
    hp = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
    dp = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    best = None
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dp.copy_(hp, non_blocking=True); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    pcie = (1 << 30) / best / 1e9
    res["pinned_copy_GBps"] = pcie
    del hp, dp
    d = tempfile.mkdtemp(prefix="dfdb_cold_", dir=base)
    try:
        path = os.path.join(d, "tb")
        t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
        t.add_generated("i", dfdb.GEN_I64_IOTA, 0, rows)
        t.add_generated("x", dfdb.GEN_I64_MOD1M, SEED, rows)
        t.add_generated("b", dfdb.GEN_I64_MOD1M, seed_of(1), rows)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = t.save(path)
        res["save"] = {"seconds": time.perf_counter() - t0, "file_GB": st["compressed"] / 1e9, "body_GB": st["uncompressed"] / 1e9,
                       "encode_GBps_incl_file_write": st["uncompressed"] / (time.perf_counter() - t0) / 1e9}
        want = t[("x", lambda x: x > THRESHOLD), dfdb.ALL]._query().count()
        t.close()
        tb = dfdb.open_table(path, ctx=ctx, load=False)
        cs = {}
        for k, name in enumerate(("i", "x", "b")):
            s_ = N.SizeStats()
            N.check(lib.dfdb_table_column_stats(tb._h, k, C.byref(s_)))
            cs[name] = {"compressed": s_.compressed, "uncompressed": s_.uncompressed}
        res["columns"] = cs
        # (1) open_table: file -> pinned -> HBM -> K7, one column
        best = None
        for _ in range(2):
            t2 = dfdb.open_table(path, ctx=ctx, load=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            t2.load(["x"]); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            t2.close()
        res["open_table"] = {"seconds": best, "rows_per_s": rows / best, "file_GBps": cs["x"]["compressed"] / best / 1e9, "decoded_GBps": rows * 8 / best / 1e9,
                             "frac_of_pinned_copy": cs["x"]["compressed"] / best / 1e9 / pcie, "what": "dfdb_table_load of column x (file in page cache) until it is resident and decoded"}

        def streamed(view, materialize_cols, reps=2):
            """the whole stream consumed through the C ABI; per chunk count (+ materialize into device buffers)"""
            q = view._query()
            nrows_chunk = chunk_blocks * 65536
            bufs = [torch.empty(nrows_chunk, dtype=torch.int64, device=dev) for _ in range(materialize_cols)]
            outs = (N.OutCol * max(materialize_cols, 1))()
            for k in range(materialize_cols):
                outs[k].data, outs[k].memkind = bufs[k].data_ptr(), N.MEM_DEVICE
            best, total, rd = None, 0, None
            for _ in range(reps):
                sh = C.c_void_p()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                N.check(lib.dfdb_stream_open(q._h, chunk_blocks, C.byref(sh)))
                total = 0
                try:
                    while True:
                        h, nr, fr = C.c_void_p(), C.c_int64(), C.c_int64()
                        N.check(lib.dfdb_stream_next(sh, C.byref(h), C.byref(nr), C.byref(fr)))
                        if not h:
                            break
                        if materialize_cols:
                            N.check(lib.dfdb_query_hint_materialize(h, 1))
                        c = C.c_int64()
                        N.check(lib.dfdb_count(h, C.byref(c)))
                        total += c.value
                        if materialize_cols and c.value:
                            N.check(lib.dfdb_materialize(h, outs, materialize_cols))
                        rd = {}
                        for k, name in enumerate(("i", "x", "b")):
                            s_ = N.SizeStats()
                            N.check(lib.dfdb_stream_read_stats(sh, k, C.byref(s_)))
                            rd[name] = s_.compressed
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                finally:
                    N.check(lib.dfdb_stream_close(sh))
                best = dt if best is None else min(best, dt)
            return best, total, rd

        def rec(sec, total, rd, what):
            file_b = sum(rd.values())
            return {"seconds": sec, "rows_per_s": rows / sec, "selected": total, "file_bytes_read": rd, "file_GBps": file_b / sec / 1e9,
                    "frac_of_pinned_copy": file_b / sec / 1e9 / pcie, "what": what}
        v = tb[("x", lambda x: x > THRESHOLD), dfdb.ALL]
        sec, total, rd = streamed(v[dfdb.ALL, ["x"]], 0, reps=3)          # (the first pass makes the slots' contexts, pinned rings and loader threads)
        res["stream_count"] = rec(sec, total, rd, "count(x > 899999), nothing resident: file -> pinned -> HBM -> K7 -> K1, chunks on four slots")
        res["stream_count"]["count_ok"] = total == want
        sec, total, rd = streamed(v[dfdb.ALL, ["x", "b"]], 2)
        res["stream_materialize"] = rec(sec, total, rd, "materialize [x, b] of x > 899999 into device buffers chunk by chunk; every block keeps rows, so b is read whole")
        thr = int(0.9 * rows)
        vc = tb[("i", lambda i: i > thr), dfdb.ALL]
        sec, total, rd = streamed(vc[dfdb.ALL, ["i", "b"]], 2)
        r = rec(sec, total, rd, "materialize [i, b] of i > 0.9 n (clustered): b is read only for the blocks with survivors (late materialization)")
        r["projection_bytes_read_frac"] = rd["b"] / cs["b"]["compressed"]
        r["count_ok"] = total == rows - thr
        res["stream_clustered"] = r
        ctx.set_option("stream_late_materialize", 0)
        try:
            sec, total, rd = streamed(vc[dfdb.ALL, ["i", "b"]], 2, reps=1)
        finally:
            ctx.set_option("stream_late_materialize", 1)
        r = rec(sec, total, rd, "the same with ctx option stream_late_materialize = 0: every required column of every chunk whole")
        r["projection_bytes_read_frac"] = rd["b"] / cs["b"]["compressed"]
        res["stream_clustered_eager"] = r

        # ---- round 6: the same table through the ORDINARY entry points (csrc/ooc.cpp): nothing is resident, dfdb_count / dfdb_materialize stream inside the library
        def ooc(view, materialize_cols, reps=2):
            ctx.set_option("ooc_chunk_blocks", chunk_blocks)
            best, total, rd = None, 0, 0
            for _ in range(reps):
                q = dfdb.api._Query(view)                    # a fresh handle per repetition: nothing counted, nothing read yet
                st0 = N.SizeStats()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                if materialize_cols:
                    q.hint_materialize(True)
                total = q.count()
                if materialize_cols:
                    bufs = [torch.empty(max(total, 1), dtype=torch.int64, device=dev) for _ in range(materialize_cols)]
                    outs = (N.OutCol * materialize_cols)()
                    for k in range(materialize_cols):
                        outs[k].data, outs[k].memkind = bufs[k].data_ptr(), N.MEM_DEVICE
                    N.check(lib.dfdb_materialize(q._h, outs, materialize_cols))
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                N.check(lib.dfdb_query_read_stats(q._h, C.byref(st0)))
                rd = st0.compressed
                best = dt if best is None else min(best, dt)
            return best, total, rd

        def rec2(sec, total, rd, what):
            return {"seconds": sec, "rows_per_s": rows / sec, "selected": total, "file_bytes_read": rd, "file_GBps": rd / sec / 1e9, "frac_of_pinned_copy": rd / sec / 1e9 / pcie, "what": what}
        sec, total, rd = ooc(v[dfdb.ALL, ["x"]], 0, reps=3)
        res["ooc_count"] = rec2(sec, total, rd, "dfdb_count(x > 899999) on the table that is not resident: ONE call, the library streams column x")
        res["ooc_count"]["count_ok"] = total == want
        sec, total, rd = ooc(v[dfdb.ALL, ["x", "b"]], 2)
        res["ooc_materialize"] = rec2(sec, total, rd, "dfdb_count + dfdb_materialize [x, b] into device buffers sized by the count: two passes like the reference's (count pre-pass, then append); "
                                                         "file bytes of both passes")
        tb.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return res


# ------------------------------------------------------------------ the line's last key: every leg in a few bytes
def _r(x, nd=3):
    return None if x is None else round(float(x), nd)


def make_summary(res):
    """`summary`, the LAST key of the line (the driver's record keeps the line's tail): per leg [ms per step or per launch, fraction of the 8 TB/s HBM peak] —
    the BASELINE configs, the decode-inclusive steps, the expression / unique / groupreduce / nullable legs — and the cold path's file GB/s with its
    fraction of the pinned-copy rate.  A leg that failed carries its error text; a leg that did not run is absent."""
    s = {"fmt": "[ms, frac_of_8TBps_peak]", "2": [_r(res.get("ms_per_step")), _r((res.get("job_hbm_gbps") or 0) / HBM_PEAK_GBPS)]}
    rf = res.get("roofline") or {}
    s["2_k1"] = [_r(rf.get("avg_launch_ms")), _r(rf.get("frac"))]
    if rf.get("box_read_ceiling_GBps"):
        s["2_k1_box"] = {"read_ceiling_GBps": _r(rf["box_read_ceiling_GBps"], 0), "k1_frac_of_it": _r(rf.get("frac_of_box_ceiling"))}
    cfg = res.get("configs") or {}
    for k in ("3", "3_computed", "4", "4_dictionary", "5_shard", "5_shard_materialize", "5_shard_dictionary", "nullable_string_eq"):
        v = cfg.get(k)
        if isinstance(v, dict):
            s[k] = v["error"][:80] if "error" in v else [_r(v.get("ms_per_step")), _r((v.get("roofline") or {}).get("frac"))]
    for k in ("unique", "unique_hash_table", "unique_float_key", "unique_string", "groupreduce", "groupreduce_int_key", "groupreduce_50k_groups", "groupreduce_hot_key", "groupreduce_float_key", "groupreduce_dictionary"):
        v = cfg.get(k)
        if isinstance(v, dict):
            s[k] = v["error"][:80] if "error" in v else [_r(v.get("seconds", 0) * 1e3), _r((v.get("roofline") or {}).get("frac"))]
    it = cfg.get("interp")
    if isinstance(it, dict):
        d = {}
        for name, v in it.items():
            if isinstance(v, dict) and ("compiled" in v or "interpreter" in v):
                d[name] = {t[:1]: [_r(v[t].get("ms")), _r(v[t].get("frac_of_peak"))] for t in ("interpreter", "compiled") if isinstance(v.get(t), dict) and "ms" in v[t]}
        s["interp"] = d if d else str(it.get("error", ""))[:80]
    for k in cfg:
        if isinstance(cfg[k], dict) and "error" in cfg[k] and k not in s:
            s[k] = cfg[k]["error"][:80]
    ds = res.get("decode_scan")
    if isinstance(ds, dict):
        if "error" in ds or "skipped" in ds:
            s["decode_scan"] = str(ds.get("error") or ds.get("skipped"))[:80]
        else:
            d = {"fmt": "[ms_per_step, K7_decoded_GBps]", "fused": [_r(ds.get("ms_per_step")), _r(ds.get("decoded_GBps"), 0)]}
            for k in ("unfused", "without_index", "arena"):
                if isinstance(ds.get(k), dict):
                    d[k] = [_r(ds[k].get("ms_per_step")), _r(ds[k].get("decoded_GBps"), 0)]
            if isinstance(ds.get("arena"), dict):
                d["arena_resident_GB"] = _r(ds["arena"].get("resident_GB"))
            s["decode_scan"] = d
    cold = res.get("cold")
    if isinstance(cold, dict):
        if "error" in cold or "skipped" in cold:
            s["cold"] = str(cold.get("error") or cold.get("skipped"))[:80]
        else:
            d = {"fmt": "[file_GBps, frac_of_pinned_copy]", "pinned_copy_GBps": _r(cold.get("pinned_copy_GBps"), 1)}
            for k in ("open_table", "stream_count", "stream_materialize", "stream_clustered", "ooc_count", "ooc_materialize"):
                if isinstance(cold.get(k), dict):
                    d[k] = [_r(cold[k].get("file_GBps"), 1), _r(cold[k].get("frac_of_pinned_copy"))]
            if isinstance(cold.get("stream_clustered"), dict):
                d["clustered_projection_read_frac"] = _r(cold["stream_clustered"].get("projection_bytes_read_frac"))
            s["cold"] = d
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("value"):
        s["cpu_1core_rows_per_s"] = _r(cb["value"], 0)
        if isinstance(ds, dict) and ds.get("rows_per_s"):
            s["decode_scan_vs_cpu_1core"] = _r(ds["rows_per_s"] / cb["value"], 1)
    return s


def finish_line(res, keep_what):
    """the line as printed: the legs' prose (`what`, `sample` stays) dropped unless --what asks for it — they were most of a 20-KB line — and `summary` LAST"""
    def strip(x):
        if isinstance(x, dict):
            return {k: strip(v) for k, v in x.items() if k != "what"}
        if isinstance(x, list):
            return [strip(v) for v in x]
        return x
    out = res if keep_what else strip(res)
    out.pop("summary", None)
    out["summary"] = make_summary(out)
    return out


# ------------------------------------------------------------------ BASELINE.json configs 3 / 4 / 5 (extra keys, never part of `value`)
def claim_stdout():
    """This script's stdout is ONE JSON line.  RCCL prints a version banner on stdout from C code the first time a communicator is used (sys.stdout
    redirection does not catch it), so file descriptor 1 points at stderr for the whole run and the JSON line is written to the descriptor saved here."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return keep


class Legs:
    """shared plumbing of the config legs: every rank runs the same leg on its own shard; a leg's time is the MAX over ranks of the wall time of
    `steps` back-to-back steps (barrier + device sync on both sides, like the headline); per-kernel times are rank 0's HIP-event averages"""

    def __init__(self, torch, dist, dev, ctx, world, rank, backend, steps, peak, calibrate=0):
        self.torch, self.dist, self.dev, self.ctx, self.world, self.rank, self.backend, self.steps, self.peak = torch, dist, dev, ctx, world, rank, backend, steps, peak
        self.calibrate = calibrate                          # ctx option placement_calibrate of the legs' contexts (the headline's setting)
        self.shards = 1                                     # --mode threads: the GPUs this ONE process drives (rows_per_s counts every shard)
        self.group, self.devices = None, None               # --mode threads: the library group whose streams a barrier must drain, its device ordinals

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        if self.group is not None:
            self.group.synchronize()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, sec):
        if self.world == 1:
            return sec
        t = self.torch.tensor([sec], dtype=self.torch.float64, device=self.dev)
        if self.backend == "nccl":
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        else:
            h = t.cpu(); self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX); t.copy_(h)
        return float(t.item())

    def timed(self, step, ctx=None):
        ctx = ctx or self.ctx
        step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(self.steps):
            step()
        self.barrier()
        sec = self.max_over_ranks((time.perf_counter() - t0) / self.steps)
        ctx.profile(True)                                 # a second, profiled pass for the per-kernel averages (event pairs on the launch stream)
        for _ in range(2):
            step()
        self.torch.cuda.synchronize()
        ks = {}
        for k in KERNELS:
            n, ms = ctx.profile_get(k)
            if n:
                ks[k] = round(ms / n, 4)
        ctx.profile(False)
        return sec, ks

    def record(self, rows, selected, sec, ks, bytes_per_gpu, what, bytes_read=None, **extra):
        """bytes_per_gpu: SURVEY.md section 8d's algorithmic bytes of the job.  A leg that reads a SMALLER representation than the one those bytes are counted on
        (dictionary codes instead of the flat String column) passes bytes_read = what it really moves: `roofline.frac` is then that over time over peak — a
        fraction of something the HBM delivered — and the flat-column figure is kept beside it as `frac_flat_equivalent`."""
        gbps = bytes_per_gpu / sec / 1e9
        r = {"what": what, "rows_per_gpu": rows, "selected_per_gpu": selected, "ms_per_step": sec * 1e3, "steps": self.steps, "kernels_avg_ms": ks,
             "algorithmic_GB": bytes_per_gpu / 1e9, "rows_per_s": rows * self.world * self.shards / sec, "placement_calibrate": self.calibrate,
             "roofline": {"bound": "hbm", "achieved": gbps, "peak": self.peak, "unit": "GB/s", "frac": gbps / self.peak,
                          "what": "algorithmic bytes of the whole job per GPU (SURVEY.md section 8d) / step time"}}
        if bytes_read is not None:
            rg = bytes_read / sec / 1e9
            r["bytes_read_GB"] = bytes_read / 1e9
            r["roofline"] = {"bound": "hbm", "achieved": rg, "peak": self.peak, "unit": "GB/s", "frac": rg / self.peak, "frac_flat_equivalent": gbps / self.peak,
                             "what": "bytes this leg really reads and writes per GPU (the dictionary's 2-byte codes stand in for the flat String column) / step time; "
                                     "frac_flat_equivalent prices the FLAT column's algorithmic bytes instead and is not a fraction of delivered bandwidth"}
        r.update(extra)
        return r


def config3_leg(L, dfdb, rows, rank):
    """config 3: 3-column Int64 + Float64 table, (a > c1) & (x < c2) at 10 %, projection [b, x] materialised into device buffers
    (docs/src/index.md:503-517: filter + projection; materialization.jl:27-40)"""
    import ctypes as C
    from dfdb import _native as N
    torch = L.torch
    t = dfdb.DFTable.new(block_size=65536, ctx=L.ctx)
    for k, (name, gen) in enumerate((("a", dfdb.GEN_I64_MOD1M), ("b", dfdb.GEN_I64_MOD1M), ("x", dfdb.GEN_F64_U2000))):
        t.add_generated(name, gen, seed_of(k), rows, row_first=rank * rows)
    q = t[(t.a > 683_771) & (t.x < 632.456), ["b", "x"]]._query()
    q.hint_materialize(True)                              # what materialize() does: the scan keeps the selected x, b is gathered
    nsel = q.count()
    ob = torch.empty(max(nsel, 1), dtype=torch.int64, device=L.dev)
    ox = torch.empty(max(nsel, 1), dtype=torch.float64, device=L.dev)
    outs = (N.OutCol * 2)()
    outs[0].data, outs[0].memkind = ob.data_ptr(), N.MEM_DEVICE
    outs[1].data, outs[1].memkind = ox.data_ptr(), N.MEM_DEVICE
    lib = N.load()

    def step():
        q.execute()
        N.check(lib.dfdb_materialize(q._h, outs, 2))
    sec, ks = L.timed(step)
    res = {"3": L.record(rows, nsel, sec, ks, rows * 16 + nsel * 8 + nsel * 16,
                         "a, b: Int64, x: Float64; (a > 683771) & (x < 632.456) -> materialize [b, x] into device buffers (x captured by the scan, b gathered)")}
    # the computed `x * 2` variant of the projection (a BlockBroadcasting column: projection.jl:128-129)
    q2 = t[(t.a > 683_771) & (t.x < 632.456), {"b": t.b, "x2": t.x * 2}]._query()
    q2.hint_materialize(True)
    assert q2.count() == nsel

    def step2():
        q2.execute()
        N.check(lib.dfdb_materialize(q2._h, outs, 2))
    sec, ks = L.timed(step2)
    res["3_computed"] = L.record(rows, nsel, sec, ks, rows * 16 + nsel * 8 + nsel * 8 + nsel * 16,
                                 "the same selection, projection [b, x * 2]: the computed column rides on a gather of x with the transform applied (k_gather_transform)")
    del ob, ox, q, q2
    t.close()
    return res


def interp_leg(L, dfdb, rows, rank):
    """Expressions no specialised scan kernel takes (arbitrary closures are the reference's normal case: it JIT-fuses every broadcast, broadcast.jl:60-68):
    the device interpreter (ctx option jit = 0) and the same program compiled at run time from the interpreter's own source by hipRTC (jit = 2: wait for
    the compiler; csrc/jit.cpp), per launch, with algorithmic GB/s = (sum of the referenced column widths + 1/8 B for the bitmap) * rows / time."""
    from dfdb import ir
    ctx = L.ctx
    t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
    t.add_generated("a", dfdb.GEN_I64_MOD1M, seed_of(0), rows, row_first=rank * rows)
    t.add_generated("b", dfdb.GEN_I64_MOD1M, seed_of(1), rows, row_first=rank * rows)
    t.add_generated("x", dfdb.GEN_F64_U2000, seed_of(2), rows, row_first=rank * rows)
    a, b, x = ir.col(0), ir.col(1), ir.col(2)
    cases = [("a*3 + b*2 - 7 > 4e6", a * 3 + b * 2 - 7 > 4_000_000, 16), ("(a > b) | (x*2 > a)", (a > b) | (x * 2 > a), 24), ("(a + b) * x > 3e9", (a + b) * x > 3e9, 24),
             ("a + b > 1.8e6", a + b > 1_800_000, 16)]
    res = {"rows_per_gpu": rows}
    for name, pred, width in cases:
        r = {"bytes_per_row": width + 0.125}
        for jit, key, kern in ((0, "interpreter", "interp_predicate"), (2, "compiled", "jit_predicate")):
            ctx.set_option("jit", jit)
            try:
                t0 = time.perf_counter()
                q = t[pred, dfdb.ALL]._query()
                r["selected"] = q.count()                                   # (jit = 2: the first execution waits for hipRTC)
                first_s = time.perf_counter() - t0
                ctx.profile(True)
                for _ in range(L.steps):
                    q.reset(); q.execute()
                L.torch.cuda.synchronize()
                n, ms = ctx.profile_get(kern)
                ctx.profile(False)
                if n:
                    gbps = rows * r["bytes_per_row"] / (ms / n * 1e-3) / 1e9
                    r[key] = {"ms": ms / n, "GBps": gbps, "frac_of_peak": gbps / L.peak, "launches": n, "first_execution_s": first_s}
                else:
                    r[key] = {"error": "the %s kernel did not run" % key}
            finally:
                ctx.set_option("jit", 1)
        res[name] = r
    t.close()
    return res


def f_rows_legs(L, dfdb, sc, rank):
    """SURVEY.md section 8(f) rows as driver-visible figures: unique over 1e9 Int64 rows with 1e6 distinct values (column.jl:102-126, docs/src/index.md:171-182),
    groupreduce by a 10-value String key (aggregate.jl:1-36), equality over a Union{String,Missing} column at config 4's size (the docs' real data set is all
    Union{Missing,String}: index.md:264-272,326-328)."""
    from dfdb import ir
    torch, ctx = L.torch, L.ctx
    res = {}
    # ---- unique
    n = int(1_000_000_000 * sc)
    t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
    t.add_generated("x", dfdb.GEN_I64_MOD1M, SEED, n, row_first=rank * n)
    best, nd = None, 0
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        u = t.x.unique()
        dt = time.perf_counter() - t0
        nd = len(u); best = dt if best is None else min(best, dt)
    res["unique"] = {"rows": n, "distinct": nd, "seconds": best, "rows_per_s": n / best, "roofline": {"bound": "hbm", "achieved": n * 8 / best / 1e9, "peak": L.peak, "unit": "GB/s", "frac": n * 8 / best / 1e9 / L.peak},
                     "what": "unique(t.x) over Int64 h mod 1e6: the keys span < 1 277 952, so a presence bit per value in LDS (one pass over the column) + the first rows from row-ordered "
                             "launches that stop once every value is found (k_unique.hip, dense form); distinct values fetched to the host in order of first appearance; best of 3"}
    ctx.set_option("unique_dense", 0)                           # the same column through the general form: the hash table that grows with the distinct values met
    try:
        best = None
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            u = t.x.unique()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    finally:
        ctx.set_option("unique_dense", 1)
    res["unique_hash_table"] = {"rows": n, "distinct": len(u), "seconds": best, "rows_per_s": n / best, "roofline": {"bound": "hbm", "achieved": n * 8 / best / 1e9, "peak": L.peak, "unit": "GB/s", "frac": n * 8 / best / 1e9 / L.peak},
                                "what": "the same unique with ctx option unique_dense = 0: the GENERAL form (what Float64 keys and wide-ranged integers take).  Round 6: for 131 K .. ~5 M distinct "
                                        "values it partitions the {key, row} records by radix and reduces each partition through a table in LDS (k_radix.hip); `hash_table_only_seconds` is the "
                                        "open-addressing table of {key, first row} in HBM it replaces there (ctx option unique_radix = 0); best of 2"}
    ctx.set_option("unique_dense", 0); ctx.set_option("unique_radix", 0)
    try:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        uh = t.x.unique()
        res["unique_hash_table"]["hash_table_only_seconds"] = time.perf_counter() - t0
        res["unique_hash_table"]["hash_table_only_distinct"] = len(uh)
    finally:
        ctx.set_option("unique_dense", 1); ctx.set_option("unique_radix", 1)
    # ---- unique over a Float64 key (x * 0.5, made on the device: 1e6 distinct values): floats always take the hash table
    t.add_column_from("f", t.x * 0.5)
    best = None
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        uf = t.f.unique()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res["unique_float_key"] = {"rows": n, "distinct": len(uf), "seconds": best, "rows_per_s": n / best, "roofline": {"bound": "hbm", "achieved": n * 8 / best / 1e9, "peak": L.peak, "unit": "GB/s", "frac": n * 8 / best / 1e9 / L.peak},
                               "what": "unique(t.f), f = x * 0.5 (Float64, 1e6 distinct values): isequal images through the general form (radix partition + LDS tables since round 6); best of 2"}
    # ---- groupreduce by a Float64 key of 5000 values ((x mod 5000) * 0.5, made on the device) over the same 1e9 rows: floats always take the hash table
    t.add_column_from("fk", (t.x % 5000) * 0.5)
    best, g = None, None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = dfdb.groupreduce(t, "fk", "x", "sum")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res["groupreduce_float_key"] = {"rows": n, "groups": len(g), "seconds": best, "rows_per_s": n / best,
                                    "roofline": {"bound": "hbm", "achieved": n * 16 / best / 1e9, "peak": L.peak, "unit": "GB/s", "frac": n * 16 / best / 1e9 / L.peak},
                                    "what": "groupreduce(t, (:fk,); out = :x => Sum()), fk = (x mod 5000) * 0.5 (Float64): the hash table is filled from the first 17 M rows (they stop bringing "
                                            "new keys), the accumulate pass looks a row's group up in a small table of the groups' keys in LDS and would report a key that is not among them; bytes = key + value columns; best of 3"}
    # ---- groupreduce by an integer key: 5000 groups (x mod 5000, made on the device) over the same 1e9 rows, sum of x
    t.add_column_from("k", t.x % 5000)
    best, g = None, None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = dfdb.groupreduce(t, "k", "x", "sum")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res["groupreduce_int_key"] = {"rows": n, "groups": len(g), "seconds": best, "rows_per_s": n / best,
                                  "roofline": {"bound": "hbm", "achieved": n * 16 / best / 1e9, "peak": L.peak, "unit": "GB/s", "frac": n * 16 / best / 1e9 / L.peak},
                                  "what": "groupreduce(t, (:k,); out = :x => Sum()), k = x mod 5000 (Int64): the keys' dense form (presence bits in LDS, no hash table) numbers the groups from the column's first 4 M rows, the "
                                          "accumulate pass adds into LDS accumulators (one 1024-thread workgroup per CU) and looks the group numbers up in an LDS copy of the table's occupied span; "
                                          "and reports a key the head did not hold (everything would then run again over every row); bytes = the key column + the value column, once each "
                                          "(24 B/row — the key column twice — while the presence pass still walked every row: 7.0 ms = 0.43 then); best of 3"}
    # ---- groupreduce over MORE groups than a workgroup's LDS accumulators hold (9216): 50 000 groups (x mod 50000), by radix since round 6 (csrc/k_radix.hip:
    # {key, row, value} records partitioned by the key's hash, a table with accumulators per partition in LDS); `atomics_seconds` is the form it replaced there
    # (every row's value to its group through a global atomic: ctx option unique_radix = 0)
    t.add_column_from("k50", t.x % 50000)
    both = {}
    for radix in (1, 0):
        ctx.set_option("unique_radix", radix)
        best, g = None, None
        try:
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                g = dfdb.groupreduce(t, "k50", "x", "sum")
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
        finally:
            ctx.set_option("unique_radix", 1)
        both[radix] = (best, len(g))
    best, ng50 = both[1]
    res["groupreduce_50k_groups"] = {"rows": n, "groups": ng50, "seconds": best, "rows_per_s": n / best, "atomics_seconds": both[0][0],
                                     "roofline": {"bound": "hbm", "achieved": n * 16 / best / 1e9, "peak": L.peak, "unit": "GB/s", "frac": n * 16 / best / 1e9 / L.peak},
                                     "what": "groupreduce(t, (:k50,); out = :x => Sum()), k50 = x mod 50000 (Int64): more groups than LDS accumulators hold — partitioned by radix "
                                             "(20-byte records) and reduced per partition in LDS; bytes = the key column + the value column, once each; best of 2"}
    # ---- the same with a HOT key: 30 % of the rows hold one key, the rest 1e5 keys evenly (through the form round 6 replaced every third row was a global atomic
    # on ONE address: 3.63 s per call, profiles/r6_groupreduce_radix.txt — not run here)
    t.add_column_from("khot", (t.x % 100000) * (t.x > 299999) + (1 << 40))
    best, g = None, None
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = dfdb.groupreduce(t, "khot", "x", "sum")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res["groupreduce_hot_key"] = {"rows": n, "groups": len(g), "largest_group_rows": int(g["count"].max()), "seconds": best, "rows_per_s": n / best,
                                  "roofline": {"bound": "hbm", "achieved": n * 16 / best / 1e9, "peak": L.peak, "unit": "GB/s", "frac": n * 16 / best / 1e9 / L.peak},
                                  "what": "groupreduce by a key that 30 % of the rows hold among 1e5 others: the partition pass reduces a hot key's rows in its own LDS slots "
                                          "(csrc/k_radix.hip); the form it replaced took 3.63 s here (global atomics on one address); best of 2"}
    t.close()
    # ---- groupreduce by a String key, flat and with the dictionary
    n = int(500_000_000 * sc)
    t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
    t.add_generated("s", dfdb.GEN_STR_BRANDS10, seed_of(0), n, row_first=rank * n)
    t.add_generated("a", dfdb.GEN_I64_MOD1M, seed_of(1), n, row_first=rank * n)
    best, us = None, None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        us = t.s.unique()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    byts = n * (4 + 5.4)
    res["unique_string"] = {"rows": n, "distinct": len(us), "seconds": best, "rows_per_s": n / best,
                            "roofline": {"bound": "hbm", "achieved": 2 * byts / best / 1e9, "peak": L.peak, "unit": "GB/s", "frac": 2 * byts / best / 1e9 / L.peak},
                            "what": "unique(t.s) over a 10-value flat String column: insert pass (hash of every string a wave has not met, {key, first row} table) + the pass that compares "
                                    "every row with its slot's representative; bytes = the key column (sizes + bytes) twice; best of 3"}
    for key in ("groupreduce", "groupreduce_dictionary"):
        if key.endswith("dictionary"):
            t.build_dictionary("s")
        best, g = None, None
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            g = dfdb.groupreduce(t, "s", "a", "sum")
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        byts = n * (4 + 5.4 + 8) if key == "groupreduce" else n * (2 + 8)
        res[key] = {"rows": n, "groups": len(g), "seconds": best, "rows_per_s": n / best,
                    "roofline": {"bound": "hbm", "achieved": byts / best / 1e9, "peak": L.peak, "unit": "GB/s", "frac": byts / best / 1e9 / L.peak},
                    "what": "groupreduce(t, (:s,); out = :a => Sum()) by a 10-value String key over Int64 values; bytes = key column (flat sizes + bytes, or 2-byte codes) + value column"}
    t.close()
    # ---- Union{String,Missing} equality at config 4's size
    t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
    t.add_generated("s", dfdb.GEN_STR_BRANDS10_MISSING, seed_of(0), n, row_first=rank * n)
    t.add_generated("a", dfdb.GEN_I64_MOD1M, seed_of(1), n, row_first=rank * n)
    q = t[ir.coalesce(ir.col(0) == "sony", False), dfdb.ALL]._query()
    nsel = q.count()
    qm = t[ir.ismissing(ir.col(0)), dfdb.ALL]._query()
    nmiss = qm.count()

    def step():
        q.reset(); q.execute()
    ctx.set_option("jit", 2)                                   # steady state: the program's compiled kernel (the default compiles it in the background)
    try:
        sec, ks = L.timed(step)
    finally:
        ctx.set_option("jit", 1)
    lbar = 5.4 * 7 / 8
    res["nullable_string_eq"] = L.record(n, nsel, sec, ks, n * (4 + lbar + 0.125), 's: Union{String,Missing} (one row in eight missing), coalesce(s == "sony", false) -> mask; '
                                         "three-valued comparison in the interpreter / its compiled kernel", missing_rows=nmiss)
    t.close()
    return res


def config4_legs(L, dfdb, rows, rank):
    """config 4: FlatStringsVector + Int64, s == "sony" (10 %), materialize s and a; flat, then with K9's dictionary codes beside the column"""
    import ctypes as C
    from dfdb import _native as N
    torch = L.torch
    t = dfdb.DFTable.new(block_size=65536, ctx=L.ctx)
    t.add_generated("s", dfdb.GEN_STR_BRANDS10, seed_of(0), rows, row_first=rank * rows)
    t.add_generated("a", dfdb.GEN_I64_MOD1M, seed_of(1), rows, row_first=rank * rows)
    q = t[t.s == "sony", dfdb.ALL]._query()
    q.hint_materialize(True)
    nsel = q.count()
    lib = N.load()
    nb = C.c_int64()
    N.check(lib.dfdb_result_string_bytes(q._h, 0, C.byref(nb)))
    osz = torch.empty(max(nsel, 1), dtype=torch.int32, device=L.dev)
    oby = torch.empty(nb.value + 64, dtype=torch.uint8, device=L.dev)
    oa = torch.empty(max(nsel, 1), dtype=torch.int64, device=L.dev)
    outs = (N.OutCol * 2)()
    outs[0].data, outs[0].bytes, outs[0].bytes_cap, outs[0].memkind = osz.data_ptr(), oby.data_ptr(), nb.value, N.MEM_DEVICE
    outs[1].data, outs[1].memkind = oa.data_ptr(), N.MEM_DEVICE

    def step():
        q.execute()
        N.check(lib.dfdb_materialize(q._h, outs, 2))
    lbar = 5.4                                            # mean brand length
    byts = rows * (4 + lbar) + nsel * (8 + 8 + 4 + 4)
    res = {}
    sec, ks = L.timed(step)
    res["4"] = L.record(rows, nsel, sec, ks, byts, 's: String (10 brands), a: Int64; s == "sony" -> materialize [s, a] into device buffers; flat FlatStringsVector scan (K5)',
                        string_bytes_out=nb.value)
    t0 = time.perf_counter()
    nd = t.build_dictionary("s")
    L.torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    sec, ks = L.timed(step)
    res["4_dictionary"] = L.record(rows, nsel, sec, ks, byts, "the same query with 16-bit dictionary codes beside s (dfdb_table_build_dictionary): the predicate is a bit-table lookup, "
                                   "s == const makes the projected s a constant column",
                                   bytes_read=rows * 2 + nsel * (8 + 8 + 4 + 4), dictionary_entries=nd, dictionary_build_s=build_s)
    del osz, oby, oa, q
    t.close()
    return res


def config5_legs(L, dfdb, G, rows, rank, local, stream, grp, host_shards):
    """config 5's per-GPU shard: mixed-type table (a Int64, x Float64, s String), conjunctive predicate over all three, count() + sum(x) — through the
    library's own group path (dfdb_group_*): every rank scans its block range and ONE exchange carries {sum, count} (docs/src/index.md:503-517,
    view.jl:192-206).  With torch.distributed over gloo (functional runs on a 1-GPU box) RCCL cannot connect ranks that share a device: the group then
    takes its exchanges from the host's collectives (dfdb_group_create_rank_callbacks) and the record says so."""
    import ctypes as C
    from dfdb import _native as N
    torch, dist, world = L.torch, L.dist, L.world
    res = {}
    total = rows * (grp.world if (grp is not None and L.shards > 1) else world)
    bytes_row = 8 + 8 + 4 + 5.4                           # a, x, sizes + bytes of s
    own = None
    if host_shards > 1:                                   # functional: ONE process, host-exchange group of `host_shards` shards on this device
        own = grp = G.Group.create([local] * host_shards, N.EXCHANGE_HOST)
        total = rows
    elif grp is None:
        if world == 1 or L.backend == "nccl":
            uid = None
            if world > 1:
                store = dist.distributed_c10d._get_default_store()
                if rank == 0:
                    store.set("dfdb_group_uid_c5", G.Group.unique_id())
                uid = bytes(store.get("dfdb_group_uid_c5"))
            own = grp = G.Group.create_rank(local, uid, rank, world, stream=stream)
        else:
            # functional runs on a 1-GPU box (--backend gloo --all-on-device0): RCCL cannot connect ranks that share a device, so the library's group
            # takes its two exchanges from the host (dfdb_group_create_rank_callbacks over torch.distributed): every other line of csrc/group.cpp is the same
            own = grp = G.Group.create_rank_torch(local, stream=stream)
    if grp is not None:
        gctx = grp.ctx(0)
        if host_shards <= 1:
            gctx.set_option("placement_calibrate", L.calibrate)
        gt = G.GroupTable.new(grp)
        gt.add_generated("a", dfdb.GEN_I64_MOD1M, seed_of(0), total)
        gt.add_generated("x", dfdb.GEN_F64_U2000, seed_of(1), total)
        gt.add_generated("s", dfdb.GEN_STR_BRANDS10, seed_of(2), total)
        v = gt.view()
        v = v[(v.a > 683_771) & (v.x < 632.456) & (v.s != "sony"), dfdb.ALL]
        gq = G.GroupQuery(gt, v[dfdb.ALL, ["x"]])
        out = {}

        def step():
            gq.reset()
            out["sum"] = gq.aggregate(N.AGG_SUM, 0)       # the scan adds x up while it holds it; ONE exchange of {sum, count}
            out["count"] = gq.count()                     # the same exchange's count: no second scan, no second collective
        exch = ("libdfdb_hip's RCCL communicator (dfdb_group_aggregate: one grouped all-reduce of {sum, count} + the fault key)" if grp.exchange == N.EXCHANGE_RCCL
                else (f"libdfdb_hip's group over the host's collectives (dfdb_group_create_rank_callbacks, torch.distributed {L.backend}; functional: the ranks share a device)"
                      if grp.exchange == N.EXCHANGE_CALLBACK else f"host exchange between {host_shards} shards of one process (functional)"))
        sec, ks = L.timed(step, gctx)
        mine = gq.shard_counts()[grp.first_rank] if host_shards <= 1 else out["count"]
        per_gpu_rows = rows if host_shards <= 1 else total
        res["5_shard"] = L.record(per_gpu_rows, mine, sec, ks, per_gpu_rows * bytes_row,
                                  "a: Int64, x: Float64, s: String; (a > 683771) & (x < 632.456) & (s != \"sony\") -> count() + sum(x), flat String scan", exchange=exch,
                                  global_count=out["count"], global_sum_x=out["sum"], total_rows=total)
        # [a, x] of the selected rows left sharded on the devices (dfdb_group_materialize_device): no PCIe, no xGMI
        gm = G.GroupQuery(gt, v[dfdb.ALL, ["a", "x"]])
        N.check(N.load().dfdb_group_query_hint_materialize(gm._h, 1))
        cnts = gm.shard_counts()[grp.first_rank:grp.first_rank + grp.nlocal]
        devs = [torch.device("cuda", d) for d in L.devices] if L.devices else [L.dev] * len(cnts)      # (every shard's outputs in ITS OWN HBM)
        bufs = [[torch.empty(max(c, 1), dtype=torch.int64, device=devs[l]), torch.empty(max(c, 1), dtype=torch.float64, device=devs[l])] for l, c in enumerate(cnts)]
        mouts = (N.OutCol * (2 * grp.nlocal))()
        for l in range(grp.nlocal):
            for p_ in range(2):
                mouts[2 * l + p_].data, mouts[2 * l + p_].memkind = bufs[l][p_].data_ptr(), N.MEM_DEVICE

        def step_m():
            gm.reset()
            N.check(N.load().dfdb_group_materialize_device(gm._h, mouts, 2))
        sec, ks = L.timed(step_m, gctx)
        nsel_m = sum(cnts) // (grp.nlocal if L.shards > 1 else 1)          # per GPU
        res["5_shard_materialize"] = L.record(per_gpu_rows, nsel_m, sec, ks, per_gpu_rows * bytes_row + nsel_m * 32,
                                              "the same selection, materialize [a, x] into per-shard DEVICE buffers (dfdb_group_materialize_device): results stay sharded", exchange=exch)
        del bufs
        gm.close()
        # the same with K9's codes beside s on every shard
        t0 = time.perf_counter()
        nd = [gt.shard(l).build_dictionary("s") for l in range(grp.nlocal)]
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0
        sec, ks = L.timed(step, gctx)
        res["5_shard_dictionary"] = L.record(per_gpu_rows, mine, sec, ks, per_gpu_rows * bytes_row,
                                             "the same with 16-bit dictionary codes beside s: s != \"sony\" scans 2 B/row", exchange=exch, bytes_read=per_gpu_rows * 18,
                                             global_count=out["count"], global_sum_x=out["sum"], dictionary_entries=nd[0], dictionary_build_s=build_s)
        gq.close(); gt.close()
        if own is not None:
            own.close()
        return res
    raise RuntimeError("unreachable: every configuration has a group")


def run_config_legs(L, dfdb, G, args, rank, local, stream, grp, out):
    sc = args.config_scale
    legs = [("3", lambda: config3_leg(L, dfdb, int(1_000_000_000 * sc), rank)),
            ("4", lambda: config4_legs(L, dfdb, int(500_000_000 * sc), rank)),
            ("5_shard", lambda: config5_legs(L, dfdb, G, int(1_250_000_000 * sc), rank, local, stream, grp, args.config5_host_shards)),
            ("interp", lambda: {"interp": interp_leg(L, dfdb, int(1_000_000_000 * sc), rank)}),
            ("f_rows", lambda: f_rows_legs(L, dfdb, sc, rank))]
    for name, fn in legs:
        try:
            out.update(fn())
        except Exception as e:      # extra figures: the headline line survives a failing leg and says which one failed.  A failure every rank meets alike
            out[name] = {"error": f"{type(e).__name__}: {e}"}      # (a missing library, a refused option) lets all of them go on; one that leaves a rank alone in an exchange ends in the deadline
    return out


def main_threads(args, out_fd):
    """--mode threads: ONE process drives N GPUs through the library's own group (dfdb_group_create: a host worker thread per GPU issues that shard's launches,
    ncclCommInitAll connects them, the count is one grouped RCCL all-reduce on the shards' engine streams) — what a Julia session calling the drop-in gets
    (INTEGRATION.md), and the path torch.distributed.run never exercises.  --all-on-device0: every shard on device 0 with the host exchange (a 1-GPU box)."""
    import torch
    torch.cuda.init()                                      # torch's copy of the HIP runtime opens the GPUs before the engine's does
    import dfdb
    from dfdb import group as G, _native as N
    n = args.gpus
    devices = [0] * n if args.all_on_device0 else list(range(n))
    grp = G.Group.create(devices, N.EXCHANGE_HOST if (args.all_on_device0 and n > 1) else N.EXCHANGE_AUTO)
    rows = args.rows
    total_rows = rows * n
    gt = G.GroupTable.new(grp)
    gt.add_generated("x", dfdb.GEN_I64_MOD1M, SEED, total_rows)
    gq = G.GroupQuery(gt, gt.view()[("x", lambda x: x > THRESHOLD), dfdb.ALL])
    cnts = gq.shard_counts()
    outs = [torch.empty(max(c, 1), dtype=torch.int64, device=torch.device("cuda", devices[l])) for l, c in enumerate(cnts)]
    ptrs, caps = [o.data_ptr() for o in outs], list(cnts)
    ctx0 = grp.ctx(0)
    info = ctx0.device_info()
    peak = float(info.get("peak_hbm_gbps") or HBM_PEAK_GBPS)

    def step():
        gq.reset()
        gq.indices_device(ptrs, caps)                      # every shard: K1 scan -> bitmap + tile counts, count scan, K2 compaction (its own worker thread)
        gq.count_async()                                   # one grouped all-reduce of the count on the engine streams, no host wait
    for _ in range(args.warmup):
        step()
    ctx0.profile(True)
    grp.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    grp.synchronize()
    elapsed = time.perf_counter() - t0
    total_sel = gq.count()
    kernels = {}
    for k in ("scan_cmp", "scan_counts", "compact_indices"):
        nl, ms = ctx0.profile_get(k)
        if nl:
            kernels[k] = dict(launches=nl, avg_ms=ms / nl)
    ctx0.profile(False)
    local_rows = gt.shard(0).view()._query().count()
    scan_row_bytes = 8 + 1 / 8 + 4 / 1024
    scan_ms = kernels.get("scan_cmp", {}).get("avg_ms")
    achieved = local_rows * scan_row_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms else None
    sigma = cnts[0] / max(local_rows, 1)
    res = {
        "metric": "filtered-scan rows/sec + achieved HBM GB/s, 1e9-row Int64 col, 10% selectivity",
        "value": total_rows * args.steps / elapsed, "unit": "rows/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": "Int64 column, selection(x -> x > 899999) -> ascending 1-based Int64 row indices + count", "rows_per_gpu": rows, "selected_per_gpu": cnts[0],
                   "selectivity": sigma, "block_size": 65536, "pipeline": "k_scan_cmp + count scan + k_compact_indices",
                   "sharding": f"contiguous block ranges x{n}, ONE process: dfdb_group_create (a host thread per GPU), count all-reduced by " +
                               ("libdfdb_hip's RCCL communicator (ncclCommInitAll)" if grp.exchange == N.EXCHANGE_RCCL else "the host exchange (shards share a device: functional)"),
                   "launcher": "one process, a host thread per GPU (--mode threads)", "device": info["name"], "global_selected": total_sel,
                   "options": "library defaults (no ctx option set; placement_calibrate = 0)", "placement_calibration": "off"},
        "job_hbm_gbps": total_rows * (8 + 8 * sigma) / (elapsed / args.steps) / 1e9,
        "roofline": {"bound": "hbm", "kernel": "k_scan_cmp<int64,GT>", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": (achieved / peak) if achieved else None,
                     "traffic": None, "traffic_source": None, "algorithmic_bytes_per_launch": local_rows * scan_row_bytes, "avg_launch_ms": scan_ms, "kernels": kernels,
                     "what": "shard 0's kernel (every shard runs the same launch on its own GPU)"},
    }
    del outs
    gq.close(); gt.close()
    if not args.no_configs:
        try:
            L = Legs(torch, None, torch.device("cuda", devices[0]), ctx0, 1, 0, "none", args.config_steps or max(3, min(args.steps, 10)), peak, 0)
            L.shards = n; L.group = grp; L.devices = devices
            res["configs"] = config5_legs(L, dfdb, G, int(1_250_000_000 * args.config_scale), 0, devices[0], None, grp, 0)
        except Exception as e:
            res["configs"] = {"5_shard": {"error": f"{type(e).__name__}: {e}"}}
    sys.stdout.flush()
    os.write(out_fd, (json.dumps(finish_line(res, args.what)) + "\n").encode())
    grp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="rows per GPU")
    ap.add_argument("--cpu-rows", type=int, default=100_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-decode-leg", action="store_true", help="skip the decode-inclusive extra figure (N = 1 only)")
    ap.add_argument("--no-cold", action="store_true", help="skip the non-resident (open_table / block-streamed) extra figures (N = 1 only)")
    ap.add_argument("--cold-rows", type=int, default=1_000_000_000, help="rows of the three-column table the `cold` leg writes to /dev/shm and streams")
    ap.add_argument("--cold-chunk-blocks", type=int, default=1024)
    ap.add_argument("--placement", action="store_true", help="also measure the step with the opt-in placement calibration (ctx option placement_calibrate = 1) and report it as the "
                    "extra key `calibrated_config`; `value` is always the library-default configuration")
    ap.add_argument("--no-placement", action="store_true", help="(accepted for older command lines: the calibration is off unless --placement asks for it)")
    ap.add_argument("--compact-store", type=int, default=None, help="A/B: K2 index stores 0 plain, 1 nontemporal, 2 write-through (ctx option compact_store)")
    ap.add_argument("--placement-column-candidates", type=int, default=None, help="A/B: fresh allocations of the column the calibration tries (ctx option placement_column_candidates; 0 = bitmaps only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL over xGMI); gloo only for functional checks")
    ap.add_argument("--all-on-device0", action="store_true", help="functional check of the N-rank path on a 1-GPU box (with --backend gloo)")
    ap.add_argument("--exchange", default="torch", choices=["torch", "lib"], help="who runs the per-step count all-reduce: torch.distributed (default) "
                    "or the library's own RCCL communicator behind the C ABI (dfdb_group_create_rank + dfdb_group_count)")
    ap.add_argument("--mode", default="processes", choices=["processes", "threads"], help="processes: one process per GPU (torch.distributed.run or self-spawned ranks); "
                    "threads: ONE process driving --gpus GPUs through dfdb_group_create (a host thread per GPU, ncclCommInitAll) — what a Julia session gets")
    ap.add_argument("--what", action="store_true", help="keep every leg's prose description (`what`) in the line (off: the line stays small enough for a record that keeps its tail)")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs 3 / 4 / 5 legs (extra keys)")
    ap.add_argument("--config-scale", type=float, default=1.0, help="scale the rows of the config legs (functional runs; 1.0 = BASELINE.json's sizes)")
    ap.add_argument("--config-steps", type=int, default=None, help="timed steps per config leg (default: min(steps, 10), at least 3)")
    ap.add_argument("--config-deadline", type=float, default=420.0, help="seconds the config legs may take before every rank ends its own process (rank 0 prints the line first)")
    ap.add_argument("--config5-host-shards", type=int, default=0, help="N = 1, functional: run the config-5 leg through a ONE-process host-exchange group of this many "
                    "shards on the device (every path of csrc/group.cpp but the RCCL calls)")
    args = ap.parse_args()

    if args.mode == "threads":
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            print("bench.py: --mode threads is one process; do not start it under torch.distributed.run", file=sys.stderr)
            sys.exit(2)
        return main_threads(args, claim_stdout())
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    out_fd = claim_stdout()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-GPU number as {args.gpus} GPUs", file=sys.stderr)
        sys.exit(2)
    if args.all_on_device0:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)

    def all_reduce(tensor, op=None):
        """the only exchange of a step: 8 bytes.  RCCL reduces the device tensor in place on the engine's stream."""
        op = op or dist.ReduceOp.SUM
        if args.backend == "nccl":
            dist.all_reduce(tensor, op=op)
        else:                                             # gloo functional path: through the host
            h = tensor.cpu()
            dist.all_reduce(h, op=op)
            tensor.copy_(h)

    import dfdb
    from dfdb import group as G
    # ONE stream for the engine's kernels, torch's tensors and the collective: a non-default torch stream made current
    # (the default stream's handle is 0, which the C ABI reads as "create a private stream")
    stream_obj = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream_obj)
    rows = args.rows
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    lib = args.exchange == "lib"
    if lib:
        # the C-ABI group: this process is rank `rank` of `world`; the RCCL id travels through torch.distributed's store
        uid = None
        if world > 1 and args.backend == "nccl":
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("dfdb_group_uid", G.Group.unique_id())
            uid = bytes(store.get("dfdb_group_uid"))
        grp = (G.Group.create_rank(local, uid, rank, world, stream=stream_obj.cuda_stream) if world == 1 or args.backend == "nccl"
               else G.Group.create_rank_torch(local, stream=stream_obj.cuda_stream))
        ctx = grp.ctx(0)
        ctx.set_option("placement_calibrate", 1 if args.placement else 0)
        nblocks_per = -(-(-(-(rows * world) // 65536)) // world)        # ceil(ceil(total / 65536) / world): the library's block-range rule
        gt = G.GroupTable.new(grp)
        gt.add_generated("x", dfdb.GEN_I64_MOD1M, SEED, rows * world)   # every rank generates its own block range
        t = gt.shard(0)
        gq = G.GroupQuery(gt, gt.view()[("x", lambda x: x > THRESHOLD), dfdb.ALL])
        nsel = gq.shard_counts()[rank]
        local_rows = t.view()._query().count()
        assert local_rows <= nblocks_per * 65536
    else:
        ctx = dfdb.Context(local, stream=stream_obj.cuda_stream)
        if args.compact_store is not None:
            ctx.set_option("compact_store", args.compact_store)
        if args.placement_column_candidates is not None:
            ctx.set_option("placement_column_candidates", args.placement_column_candidates)
        t = dfdb.DFTable.new(block_size=65536, ctx=ctx)
        t.add_generated("x", dfdb.GEN_I64_MOD1M, SEED, rows, row_first=rank * rows)   # this rank's block range
        t.set_row_base(rank * rows)
        q = t[("x", lambda x: x > THRESHOLD), dfdb.ALL]._query()       # library defaults: no calibration, the allocations as hipMalloc handed them out
        nsel = q.count()                                   # exact selected count from the first (untimed) execution
        local_rows = rows
    info = ctx.device_info()
    cap = nsel
    out = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)

    if lib:
        def step():
            gq.reset()                                     # a fresh evaluation every step (nothing cached)
            gq.indices_device([out.data_ptr()], [cap])     # K1 scan -> bitmap + tile counts, count scan, K2 compaction
            gq.count_async()                               # all-reduce of the count on the engine stream, no host wait
    else:
        def step():
            q.reset()
            q.indices_device(out.data_ptr(), cap)
            q.count_device(cnt.data_ptr())
            if world > 1:
                all_reduce(cnt)

    peak = float(info.get("peak_hbm_gbps") or HBM_PEAK_GBPS)      # dfdb_ctx_device_info: 8000 on MI355X
    scan_row_bytes = 8 + 1 / 8 + 4 / 1024                          # ONE k_scan_cmp launch: 8 B/row column + 1/8 B/row bitmap + 4 B per 1024-row tile count
    total_rows = rows * world

    for _ in range(args.warmup):
        step()
    # what this box / this allocation gives K1's read stream with nothing written beside it (dfdb_table_read_probe: the scan's load shape, no stores): the
    # ceiling K1's own rate is read against — a slow box or an unlucky placement shows here, a slower kernel does not
    probe_best_ms = probe_avg_ms = None
    try:
        probe_best_ms, probe_avg_ms = t.read_probe("x", 10)
    except Exception as e:
        print(f"bench.py: read probe failed: {e}", file=sys.stderr)
    step()
    # roofline leg: HIP event pairs around every launch, recorded on the launch stream DURING the timed steps and resolved
    # after them (no host synchronisation inside the region: dfdb_ctx_profile_get folds them)
    ctx.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce(el, dist.ReduceOp.MAX)
        elapsed = float(el.item())
    total_sel = gq.count() if lib else int(cnt.item())

    kernels = {}
    for k in ("scan_cmp", "scan_counts", "compact_indices"):
        n, ms = ctx.profile_get(k)
        if n:
            kernels[k] = dict(launches=n, avg_ms=ms / n)
    ctx.profile(False)

    # ---- opt-in (--placement): the same step after the engine's one-time placement calibration (query.cpp: place_mask re-places the column and picks a
    # bitmap allocation).  An extra key; `value` above is what a caller gets who sets no option.
    calibrated_cfg = None
    if args.placement and not lib:
        ctx.set_option("placement_calibrate", 1)
        q = t[("x", lambda x: x > THRESHOLD), dfdb.ALL]._query()      # a fresh query's first execution runs the calibration
        assert q.count() == nsel
        for _ in range(max(args.warmup, 1)):
            step()
        ctx.profile(True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el2 = time.perf_counter() - t0
        if world > 1:
            e2 = torch.tensor([el2], dtype=torch.float64, device=dev)
            all_reduce(e2, dist.ReduceOp.MAX)
            el2 = float(e2.item())
        n2, ms2 = ctx.profile_get("scan_cmp")
        n3, ms3 = ctx.profile_get("compact_indices")
        ctx.profile(False)
        sms = ms2 / n2 if n2 else None
        calibrated_cfg = {"what": "the same step with ctx option placement_calibrate = 1 (opt-in), measured after `value`",
                          "value": total_rows * args.steps / el2, "ms_per_step": el2 / args.steps * 1e3, "scan_cmp_avg_ms": sms, "compact_indices_avg_ms": ms3 / n3 if n3 else None,
                          "roofline_frac": (local_rows * scan_row_bytes / (sms * 1e-3) / 1e9 / peak) if sms else None,
                          "job_hbm_gbps": total_rows * (8 + 8 * nsel / local_rows) / (el2 / args.steps) / 1e9}
    pl_n, pl_best = ctx.profile_get("placement_best_us")
    _, pl_worst = ctx.profile_get("placement_worst_us")
    _, pl_wall = ctx.profile_get("placement_wall_us")
    pc_n, pc_best = ctx.profile_get("placement_column_best_us")
    _, pc_worst = ctx.profile_get("placement_column_worst_us")
    pt_n, pt_best = ctx.profile_get("placement_counts_best_us")
    _, pt_worst = ctx.profile_get("placement_counts_worst_us")
    res = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rows * args.steps / elapsed
        sigma = nsel / local_rows
        kname = "k_scan_cmp<int64,GT>"
        scan_bytes = local_rows * scan_row_bytes
        scan_ms = kernels.get("scan_cmp", {}).get("avg_ms")
        achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms else None
        job_bytes = total_rows * (8 + 8 * sigma)       # 8 + 8*sigma B/row for the whole job
        # HBM traffic of that kernel from the PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
        # same command; FETCH_SIZE doubled per the gfx950 correction, calibrated on a known-byte read): profiles/
        traffic, traffic_src = None, None
        import glob
        import re
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_scan_cmp.json")), key=lambda f: -int(re.search(r"r(\d+)_pmc", f).group(1)))
        for pmc in pmcs:                                   # the newest round's PMC passes first
            with open(pmc) as f:
                pj = json.load(f)
            if pj.get("rows"):
                traffic = pj["hbm_bytes_per_launch_corrected"] * local_rows / pj["rows"]
                traffic_src = f"profiles/{os.path.basename(pmc)} (2*FETCH_SIZE + WRITE_SIZE, scaled to rows)"
                break
        res = {
            "metric": "filtered-scan rows/sec + achieved HBM GB/s, 1e9-row Int64 col, 10% selectivity",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": "Int64 column, selection(x -> x > 899999) -> ascending 1-based Int64 row indices + count",
                       "rows_per_gpu": rows, "selected_per_gpu": nsel, "selectivity": sigma, "block_size": 65536,
                       "pipeline": "k_scan_cmp + count scan + k_compact_indices",
                       "sharding": (f"contiguous block ranges x{world}, all-reduce(count) per step by " +
                                    (("libdfdb_hip's RCCL communicator (dfdb_group_count)" if args.backend == "nccl" else f"libdfdb_hip's group over the host's collectives (torch.distributed {args.backend})")
                                     if lib else f"torch.distributed {args.backend}")) if world > 1 else "single GPU",
                       "launcher": "torch.distributed.run" if os.environ.get("TORCHELASTIC_RUN_ID") else ("bench.py spawned its own ranks" if world > 1 else "single process"),
                       "device": info["name"], "global_selected": total_sel,
                       "placement_calibration": ({"candidates_best_ms": pl_best / 1e3, "candidates_worst_ms": pl_worst / 1e3, "sample_rows": local_rows, "one_time_seconds": pl_wall / 1e6,
                                                  "column_candidates_best_ms": pc_best / 1e3 if pc_n else None, "column_candidates_worst_ms": pc_worst / 1e3 if pc_n else None,
                                                  **({"count_candidates_best_ms": pt_best / 1e3, "count_candidates_worst_ms": pt_worst / 1e3} if pt_n else {}),
                                                  "what": "one-time: the scan timed on fresh allocations of the column (device-to-device copies; the fastest becomes the column), then against "
                                                          "9 bitmap allocations, the fastest kept (ctx option placement_calibrate, opt-in: --placement; `value` is measured before it)"}
                                                 if pl_n else "off")},
            "job_hbm_gbps": job_bytes / (elapsed / args.steps) / 1e9,
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": peak, "unit": "GB/s",
                         "frac": (achieved / peak) if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": scan_bytes, "avg_launch_ms": scan_ms, "kernels": kernels},
        }
        if probe_best_ms:
            ceil = local_rows * 8 / (probe_best_ms * 1e-3) / 1e9
            res["roofline"].update({"box_read_ceiling_GBps": ceil, "box_read_ceiling_avg_GBps": local_rows * 8 / (probe_avg_ms * 1e-3) / 1e9,
                                    "frac_of_box_ceiling": (achieved / ceil) if achieved else None,
                                    "box_read_ceiling_source": "dfdb_table_read_probe: K1's load shape over the same resident column, no stores, best / average of 10 launches before the timed steps"})
        res["config"]["options"] = "library defaults (no ctx option set; placement_calibrate = 0)" if not (lib and args.placement) else "placement_calibrate = 1"
        if calibrated_cfg is not None:
            res["calibrated_config"] = calibrated_cfg
        if world == 1 and not lib and not args.no_decode_leg:
            try:
                res["decode_scan"] = decode_scan_leg(dfdb, ctx, t, rows, max(3, min(args.steps, 10)), out.data_ptr(), cap, cnt.data_ptr(), torch.cuda.synchronize)
            except Exception as e:      # an extra figure: never fail the bench line for it
                res["decode_scan"] = {"error": f"{type(e).__name__}: {e}"}
    def emit(line):
        sys.stdout.flush()
        os.write(out_fd, (json.dumps(finish_line(line, args.what)) + "\n").encode())

    # ---- configs 3 / 4 / 5 on every rank (the column of config 2 goes first: the legs bring their own tables)
    if not args.no_configs:
        # The legs are extra keys; the headline must survive them.  They exchange between ranks (barriers, the library's group): should a rank
        # ever be left waiting in one (the N > 1 path has only run on one physical GPU so far), a deadline ends EVERY rank's process — rank 0 prints
        # the line first, with whatever legs had finished and a note — instead of the whole run dying in the driver's timeout with nothing printed.
        import threading
        done_legs = {}

        def deadline():
            if rank == 0 and res is not None:
                res["configs"] = dict(done_legs, error=f"the config legs did not finish within {args.config_deadline} s: the process was ended by its own deadline")
                emit(res)
            os._exit(3)                                    # the line is out, and the exit status says the run did not end by itself (never a re-exec, nothing started)
        watchdog = threading.Timer(args.config_deadline, deadline)
        watchdog.daemon = True
        watchdog.start()
        calibrate = 1 if (args.placement and args.config5_host_shards <= 1) else 0   # library defaults unless --placement asks for the opt-in calibration
        ctx.set_option("placement_calibrate", calibrate)
        del out
        if not lib:
            t.close()
        torch.cuda.empty_cache()
        L = Legs(torch, dist, dev, ctx, world, rank, args.backend, args.config_steps or max(3, min(args.steps, 10)), peak, calibrate)
        legs = run_config_legs(L, dfdb, G, args, rank, local, stream_obj.cuda_stream, grp if lib else None, done_legs)
        watchdog.cancel()
        if res is not None:
            res["configs"] = legs
    if res is not None and world == 1 and not lib and not args.no_cold:
        try:
            torch.cuda.empty_cache()
            res["cold"] = cold_leg(dfdb, ctx, torch, dev, args.cold_rows, peak, args.cold_chunk_blocks)
        except Exception as e:          # an extra figure: never fail the bench line for it
            res["cold"] = {"error": f"{type(e).__name__}: {e}"}
    if res is not None:
        if not args.no_cpu and world == 1:      # the CPU baseline is an N=1, rank-0 figure
            res["cpu_baseline"] = cpu_baseline(args.cpu_rows, 3)
        emit(res)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
