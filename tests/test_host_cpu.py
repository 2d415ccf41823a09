"""CPU tests (no GPU) of the host side: IR front-end, the Python mirror's lazy algebra, the C-ABI surface
of libdfdb_hip.so (load + exported symbols, no compute), and the multi-rank planning over gloo."""
import os
import re
import socket
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ IR
def test_ir_serialisation_layout():
    from dfdb import ir
    e = (ir.col(3) > 899_999)
    b = e.to_ir()
    assert b == bytes([ir.COL]) + struct.pack("<I", 3) + bytes([ir.CONST, ir.I64]) + struct.pack("<q", 899_999) + bytes([ir.GT])
    e2 = ir.startswith(ir.col(1), "1")
    assert e2.to_ir() == bytes([ir.COL]) + struct.pack("<I", 1) + bytes([ir.CONST_STR]) + struct.pack("<I", 1) + b"1" + bytes([ir.STARTSWITH])
    e3 = ir.isin(ir.col(0), [1, 11, 21])
    assert e3.to_ir().endswith(struct.pack("<BBI", ir.CONST_SET, ir.I64, 3) + struct.pack("<qqq", 1, 11, 21) + bytes([ir.IN_SET]))
    assert (ir.col(0) * 2.5).to_ir()[5:15] == bytes([ir.CONST, ir.F64]) + struct.pack("<d", 2.5)
    assert (65 > ir.col(0)).to_ir()[-1] == ir.LT            # reflected comparison: a < 65
    assert ((ir.col(0) + ir.col(2)) / 3).columns() == [0, 2]
    with pytest.raises(TypeError):
        bool(ir.col(0) > 1)                                 # `and` / chained comparisons cannot be traced
    with pytest.raises(TypeError):
        ir.col(0) // 2


def test_tracer_and_remap():
    from dfdb import ir
    e = ir.trace(lambda a, c: (a % 50 == 0) & (c < 930), [ir.col(0), ir.col(2)])
    assert e.columns() == [0, 2] and e.op == ir.AND
    assert e.same(ir.trace(lambda a, c: (a % 50 == 0) & (c < 930), [ir.col(0), ir.col(2)]))
    assert not e.same(ir.trace(lambda a, c: (a % 50 == 0) & (c < 931), [ir.col(0), ir.col(2)]))
    nested = ir.trace(lambda k: k > 3, [ir.col(0) + ir.col(1)])       # predicate over a computed projection column (Q14)
    assert nested.columns() == [0, 1]


# ------------------------------------------------------------------ SelectionQueue / Projection (no engine needed)
def test_selection_queue_composition():
    """test/selection.jl:5-37 on the mirror."""
    from dfdb import SelectionQueue, jr, ir, ALL
    sel = SelectionQueue()
    assert sel.isempty()
    assert sel.add(ALL).isempty()
    s2 = sel.add(jr(5, 20))
    assert len(s2) == 1
    s2 = s2.add(jr(1, 5))
    assert len(s2) == 1 and s2.queue[0] == jr(5, 9)
    tb = ir.col(0) == 1
    s2 = s2.add(tb)
    assert len(s2) == 2
    s3 = sel.add(tb).add(tb)
    assert len(s3) == 1 and s3.queue[0].op == ir.AND
    s3 = s3.add(jr(1, 3))
    assert len(s3) == 2
    with pytest.raises(IndexError):
        sel.add(jr(5, 20)).add(jr(1, 30))
    assert sel.add(jr(10, 2, 30)).add(jr(2, 2, 6)).queue[0] == jr(12, 4, 20)
    assert sel.add(jr(10, 2, 30)).add([3, 1]).queue[0] == [14, 10]
    assert sel.add([5, 6, 7]).add(jr(2, 3)).queue[0] == [6, 7]
    assert sel.add(jr(5, 20)).add(3).queue[0] == 7
    with pytest.raises(IndexError):
        sel.add(7).add(2)
    assert sel.add(jr(1, 30)).same(sel.add(jr(1, 30))) and not sel.add(jr(1, 30)).same(sel.add(jr(1, 20)))


def test_projection_ops():
    """test/projection.jl:6-55 on the mirror."""
    from dfdb import Projection, ir
    p = Projection({"a": ir.col(0), "b": ir.col(0) * 2})
    assert p.keys() == ["a", "b"]
    with pytest.raises(ValueError):
        p.add({"a": ir.col(0)})
    p2 = Projection({"a": ir.col(0)}).add({"c": ir.col(1), "e": ir.col(4)})
    assert p2.keys() == ["a", "c", "e"]
    assert p2.select_positions([2]).keys() == ["c"]
    assert p2.select_positions([1, 2]).keys() == ["a", "c"]
    assert p2.select_positions([1, 3]).keys() == ["a", "e"]
    assert p2.select_names(["e", "a"]).keys() == ["a", "e"]          # projection order wins (quirk Q13)
    with pytest.raises(IndexError):
        p2.select_positions([4])
    assert len(Projection()) == 0


# ------------------------------------------------------------------ C ABI surface
def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "dfdb.h")).read()
    return sorted(set(re.findall(r"^int32_t\s+(dfdb_\w+)\s*\(", txt, re.M)))


def test_abi_exports_every_declared_symbol():
    import ctypes
    import dfdb
    declared = _header_symbols()
    assert sorted(dfdb.SYMBOLS) == declared, set(declared) ^ set(dfdb.SYMBOLS)
    lib = dfdb.load()                       # dlopen of the in-tree HIP library works without a GPU
    for s in declared:
        assert isinstance(getattr(lib, s), ctypes._CFuncPtr)
    assert lib.dfdb_version() == 1


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible")
    import dfdb
    with pytest.raises(dfdb.DfdbError, match="no CPU fallback|no HIP device"):
        dfdb.Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dataframedbs.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".jl")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                for bad in ("liboracle", "orc_", "oracle/", "import oracle", "from oracle", "oracle.h"):
                    assert bad not in txt, f"{os.path.join(dirpath, f)} mentions {bad!r}"


# ------------------------------------------------------------------ sharding
def test_block_ranges_cover_table():
    from dfdb.sharding import block_range, row_range
    for nb in (0, 1, 7, 8, 9, 15259, 152588):
        for w in (1, 2, 4, 8):
            rs = [block_range(nb, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == nb
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
    assert row_range(10**9, 65536, 7, 8) == (7 * 1908 * 65536, 10**9)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _eval_shard(a, r0, stages, bases):
    """numpy stand-in for one rank's engine: stages over local rows with global numbering."""
    mask = np.ones(len(a), bool)
    for i, st in enumerate(stages):
        if st[0] == "pred":
            mask &= st[1](a)
        else:
            if i == 0:
                rank = r0 + np.arange(1, len(a) + 1)
            else:
                rank = bases[i] + np.cumsum(mask)
            keep = np.isin(rank, np.arange(st[1], st[3] + 1, st[2])) if st[0] == "range" else np.isin(rank, st[1])
            mask &= keep
    return mask


def _worker(rank, world, port, n, bs, q):
    import torch.distributed as dist
    from dfdb.sharding import plan_stage_bases, row_range, all_reduce_scalars
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r0, r1 = row_range(n, bs, rank, world)
        a = np.arange(r0 + 1, r1 + 1, dtype=np.int64)
        results = {}
        for name, stages in _QUERIES.items():
            bases = {}
            kinds = ["predicate" if s[0] == "pred" else ("range" if s[0] == "range" else "indices") for s in stages]
            plan_stage_bases(kinds, lambda k: int(_eval_shard(a, r0, stages[:k], bases).sum()), lambda k, b: bases.__setitem__(k, b))
            m = _eval_shard(a, r0, stages, bases)
            tot = all_reduce_scalars([int(m.sum())], "sum")[0]
            results[name] = ((a[m]).tolist(), int(tot))
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


_QUERIES = {
    "pred_only": [("pred", lambda a: a % 3 == 1)],
    "lead_range": [("range", 5, 7, 90_000), ("pred", lambda a: a % 2 == 0)],
    "range_after_pred": [("pred", lambda a: a % 3 == 1), ("range", 5, 3, 20_000)],
    "two_exchanges": [("pred", lambda a: a % 2 == 0), ("range", 10, 1, 30_000), ("pred", lambda a: a % 3 == 0), ("idx", [1, 5, 4000, 9999])],
}


def test_two_rank_plan_matches_oracle(oracle):
    """world_size 2 over gloo: per-shard evaluation + the exclusive-scan exchange reproduces the oracle's
    single-table answer, including range stages that index the survivor stream (quirk Q1)."""
    import torch.multiprocessing as mp
    from dfdb import ir
    n, bs, world = 100_003, 4096, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, bs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    t = oracle.Table(block_size=bs)
    t.add_column("a", np.arange(1, n + 1, dtype=np.int64))
    A = ir.col(0)
    oq = {
        "pred_only": t.view().add_predicate((A % 3 == 1).to_ir()),
        "lead_range": t.view().add_range(5, 7, 90_000).add_predicate((A % 2 == 0).to_ir()),
        "range_after_pred": t.view().add_predicate((A % 3 == 1).to_ir()).add_range(5, 3, 20_000),
        "two_exchanges": t.view().add_predicate((A % 2 == 0).to_ir()).add_range(10, 1, 30_000).add_predicate((A % 3 == 0).to_ir()).add_indices([1, 5, 4000, 9999]),
    }
    for name, ov in oq.items():
        want = ov.select_indices().tolist()
        cat = got[0][name][0] + got[1][name][0]          # rank order = table order
        assert cat == want, name
        assert got[0][name][1] == got[1][name][1] == len(want)


def test_flat_strings_to_arrow_matches_the_per_row_conversion():
    """set_string_output("arrow") wraps the engine's (sizes, arena) pair without per-row work; it must denote the same strings,
    missing rows (size -1) included, as the Vector{String}-like object conversion (projection.jl:99-100)."""
    from dfdb import api
    rng = np.random.default_rng(3)
    words = ["", "a", "sony", "né", "microsoft", "x" * 40]
    pick = rng.integers(0, len(words), 5000)
    sizes = np.array([len(words[i].encode()) for i in pick], np.int32)
    sizes[::11] = -1
    data = np.frombuffer(b"".join(words[i].encode() for k, i in enumerate(pick) if sizes[k] >= 0), np.uint8)
    assert api._flat_to_arrow(sizes, data).to_pylist() == api._flat_to_strings(sizes, data)
    assert api._flat_to_arrow(np.zeros(0, np.int32), np.zeros(0, np.uint8)).to_pylist() == []


def test_julia_shim_ccalls_match_the_header():
    """Julia is not installed here, so the shim's FFI is checked statically: every ccall of julia/DataFrameDBsAMD.jl against the
    prototype of the same symbol in include/dfdb.h (exists, arity, argument class and width, Int32 return) and the two mirrored
    structs field by field; julia/STATIC_REVIEW.md is the committed rendering of the same walk and must be current."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("julia_static_review", os.path.join(ROOT, "tools", "julia_static_review.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows, errors = mod.check()
    assert not errors, errors
    assert len(rows) >= 50
    # the semantic lint (checked integer conversions, unrooted pointers, unchecked statuses) is clean on the shim ...
    assert mod.lint() == [], mod.lint()
    # ... and does find what it is there for: round 2's constant emitter, a bare pointer, a status nobody looks at
    import tempfile
    real = mod.JL
    with tempfile.NamedTemporaryFile("w", suffix=".jl", delete=False) as f:
        f.write("function emit_const(io, v)\n    write(io, Int64(v) % Int64)\nend\n"
                "function f(buf, q, n)\n    p = pointer(buf)\n    ccall((:dfdb_count, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), q, n)\n    n[]\nend\n")
    try:
        mod.JL = f.name
        bad = mod.lint()
    finally:
        mod.JL = real
        os.unlink(f.name)
    assert len(bad) == 3 and "Int64(v)" in bad[0] and "pointer(buf)" in bad[1] and "dfdb_count" in bad[2], bad
    # round 4: the closure-lowering rules (the walk over Base.code_lowered) are each pinned to a line of the shim and rendered into STATIC_REVIEW.md
    wrows, wmissing = mod.walker_review()
    assert not wmissing and len(wrows) >= 12 and all(at for _n, at, _w in wrows), wmissing
    syms = {r[1] for r in rows}
    # round 3: sharded unique / groupreduce go through the library's own merge
    for s in ("dfdb_group_query_unique", "dfdb_group_query_unique_fetch", "dfdb_group_query_groupreduce", "dfdb_group_query_groupreduce_fetch"):
        assert s in syms, s
    # the consumers VERDICT r1 asked for are bound: one-column routes, aggregates, unique, and the multi-GPU group
    for s in ("dfdb_materialize", "dfdb_aggregate", "dfdb_query_unique", "dfdb_group_create", "dfdb_group_count", "dfdb_group_aggregate", "dfdb_group_materialize"):
        assert s in syms, s
    # round 6: nothing is loaded when a table is opened — the shim binds neither dfdb_table_load nor dfdb_group_table_load —, every view
    # prepares exactly its required columns after its projection is set and before anything is asked of it, DFDB_ERR_NOMEM is OutOfMemoryError (not
    # ErrorException) and is answered by unloading the table and asking once more, block-streamed
    assert "dfdb_table_load" not in syms and "dfdb_group_table_load" not in syms
    for s in ("dfdb_query_prepare", "dfdb_group_query_prepare", "dfdb_table_unload", "dfdb_group_table_unload", "dfdb_query_reset", "dfdb_group_query_reset"):
        assert s in syms, s
    jl = open(mod.JL).read()
    assert "rc == 9 && throw(OutOfMemoryError())" in jl
    builder = jl[jl.index("for (fname, tabfn, NEW, FREE"):jl.index("# struct dfdb_outcol")]
    i_proj, i_prep, i_f = builder.index("QuoteNode(PROJ)"), builder.index("QuoteNode(PREPARE)"), builder.index("return f(q[])")
    i_catch, i_unload, i_reset = builder.index("e isa OutOfMemoryError || rethrow()"), builder.index("QuoteNode(UNLOAD)"), builder.index("QuoteNode(RESET)")
    assert i_proj < i_prep < i_f < i_catch < i_unload < i_reset < builder.rindex("return f(q[])")
    assert builder.count("return f(q[])") == 2                      # the streamed retry happens once
    for env in ("DFDB_HBM_BUDGET_MB", "DFDB_OOC_CHUNK_BLOCKS"):
        assert env in jl
    review = open(os.path.join(ROOT, "dataframedbs.jl_amd", "julia", "STATIC_REVIEW.md")).read()
    assert f"**{len(rows)} ccall sites, 0 mismatches**" in review, "run tools/julia_static_review.py"
    assert "## Closures with control flow: the walk over lowered code (round 4)" in review and "**MISSING**" not in review, "run tools/julia_static_review.py"
    jl = open(os.path.join(ROOT, "dataframedbs.jl_amd", "julia", "DataFrameDBsAMD.jl")).read()
    assert "Base.invoke_in_world(WORLD0[]" in jl and "invoke(DataFrameDBs._cpu" not in jl          # the fallback cannot re-enter an override
    for route in ("DataFrameDBs.materialize(c::DFColumn)", "Base.copyto!(dest::AbstractVector, src::DFColumn)", "Base.sum(c::DFColumn)",
                  "Statistics.mean(c::DFColumn)", "Base.unique(c::DFColumn)", "Broadcasted{DataFrameDBs.DFColumnStyle}"):
        assert route in jl, route


def test_sharded_unique_and_groupreduce_live_behind_the_abi():
    """round 2 merged the shards' unique / groupreduce records in Python (dfdb/group.py: _key_of, _all_ranks over torch.distributed); since round 3
    the merge is the library's (csrc/group.cpp: group_reduce_all) and the Python layer only sizes buffers — a Julia caller gets the same answers"""
    from dfdb import group as G
    import dfdb
    for gone in ("_key_of", "_all_ranks", "_shard_queries"):
        assert not hasattr(G, gone), gone
    for s in ("dfdb_group_query_unique", "dfdb_group_query_unique_fetch", "dfdb_group_query_groupreduce", "dfdb_group_query_groupreduce_fetch",
              "dfdb_group_materialize_device", "dfdb_group_shard_string_bytes"):
        assert s in dfdb.SYMBOLS
    hdr = open(os.path.join(ROOT, "include", "dfdb.h")).read()
    assert "first appearance = lowest rank, then" in hdr and "fault key" in hdr


def test_ctypes_structs_mirror_the_header():
    """every struct that crosses the ABI by value or by pointer: the Python mirror has the header's fields, in the header's order, in the header's widths"""
    import ctypes as C
    import re
    from dfdb import _native as N
    hdr = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "dfdb.h")).read(), flags=re.S)
    width = {"int32_t": 4, "int64_t": 8, "double": 8, "uint64_t": 8}
    for cname, py in (("dfdb_device_info", N.DeviceInfo), ("dfdb_colinfo", N.ColInfo), ("dfdb_sizestats", N.SizeStats), ("dfdb_outcol", N.OutCol), ("dfdb_exchange_fns", N.ExchangeFns)):
        body = re.search(r"typedef struct " + cname + r"\s*\{(.*?)\}\s*" + cname + ";", hdr, re.S).group(1)
        want = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fp = re.match(r"\w+ \(\*(\w+)\)\(", decl)                       # a function pointer member
            if fp:
                want.append((fp.group(1), C.sizeof(C.c_void_p))); continue
            base = decl.split()[0]
            for nm in decl[len(base):].split(","):
                nm = nm.strip()
                arr = re.match(r"(\w+)\[(\d+)\]", nm)
                if arr:
                    want.append((arr.group(1), int(arr.group(2)))); continue     # char name[128]
                want.append((nm.replace("*", "").strip(), C.sizeof(C.c_void_p) if "*" in nm or "*" in base else width[base]))
        got = [(n, C.sizeof(t)) for n, t in py._fields_]
        assert got == want, (cname, got, want)


def test_oracle_is_clean_under_address_and_ub_sanitizers():
    """The parity suites take the oracle's word; it must not owe its answers to undefined behaviour.  tests/oracle_sanitize_soak.py rebuilds it with
    gcc -fsanitize=address,undefined and evaluates the fuzz suite's random queues / projections (and file round trips) with the oracle alone."""
    import shutil
    import subprocess
    import sys
    if not shutil.which("gcc") or not os.path.exists(subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()):
        pytest.skip("gcc's sanitizer runtimes are not installed")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "oracle_sanitize_soak.py"), "--seeds", "1500", "--seed0", "7000000"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    assert "no sanitizer report" in r.stdout.decode()


def test_every_context_option_the_engine_reads_is_documented_in_the_header():
    """dfdb_ctx_set_option takes free-form keys: an option the engine reads must be findable — the supported ones (at most twenty: VERDICT r5 weak 10) in
    include/dfdb.h, the A/B and test knobs in csrc/KNOBS.md — and nothing is listed in either place that the engine no longer reads."""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "dataframedbs.jl_amd", "csrc", "*.[ch]pp")) + glob.glob(os.path.join(ROOT, "dataframedbs.jl_amd", "csrc", "*.hip"))
    src = "".join(open(f).read() for f in files)
    read = set(re.findall(r'ctx_option\((?:[^,()]|\([^()]*\))+,\s*"([a-z0-9_]+)"', src))
    assert len(read) >= 30, read
    header = open(os.path.join(ROOT, "include", "dfdb.h")).read()
    block = header[header.index("/* Context options (free-form keys"):header.index("int32_t dfdb_ctx_set_option(")]
    public = set(re.findall(r'^ \*   "([a-z0-9_]+)"', block, re.M))
    knobs_md = open(os.path.join(ROOT, "dataframedbs.jl_amd", "csrc", "KNOBS.md")).read()
    knobs = set(re.findall(r'^\| `([a-z0-9_]+)`', knobs_md, re.M))
    assert len(public) <= 20, sorted(public)
    assert not (public & knobs), public & knobs
    assert read - public - knobs == set(), sorted(read - public - knobs)
    assert (public | knobs) - read == set(), sorted((public | knobs) - read)


# ------------------------------------------------------------------ round 5: host-side hardening that needs no GPU
def test_a_failed_collective_always_closes_its_rccl_group_and_marks_the_communicator_dead():
    """csrc/group.cpp's ncclGroupStart / ncclGroupEnd bracket, driven by a stub collective table (dfdb_selftest "rccl_bracket"): whichever of
    the three all-reduces fails — or the bracket's own calls — GroupEnd has run as often as GroupStart succeeded, the failing exchange returns an error,
    and the NEXT exchange on that communicator fails at once without touching RCCL again (no nesting, no hang)."""
    import ctypes as C
    from dfdb import _native as N
    lib = N.load()
    out = (C.c_int64 * 6)()

    def run(fail_at):
        assert lib.dfdb_selftest(b"rccl_bracket", fail_at, out, 6) == 0
        return list(out)
    assert run(0) == [2, 2, 6, 0, 0, 0]                                    # two healthy exchanges: two brackets, six collectives
    for k in (1, 2, 3):
        starts, ends, calls, first, second, dead = run(k)
        assert (starts, ends) == (1, 1), "GroupEnd must run for the bracket whose collective failed, and no second bracket may open"
        assert calls == k and first == N.ERR_DEVICE and second == N.ERR_DEVICE and dead == 1
    assert run(-1) == [1, 0, 0, N.ERR_DEVICE, N.ERR_DEVICE, 1]             # GroupStart itself failed: nothing to close
    assert run(-2) == [1, 1, 3, N.ERR_DEVICE, N.ERR_DEVICE, 1]             # GroupEnd failed: still called exactly once
    assert lib.dfdb_selftest(b"no_such_test", 0, out, 6) == N.ERR_ARGUMENT
    assert lib.dfdb_selftest(b"rccl_bracket", 0, out, 2) == N.ERR_ARGUMENT


def test_the_jit_disk_cache_refuses_directories_it_cannot_trust(tmp_path, monkeypatch):
    """VERDICT r4 item 7 / ADVICE: a code object read from the cache runs in this process's GPU context, so the directory must be a real directory owned by the
    user and writable by nobody else; anything else turns the cache off (dfdb_jit_cache_dir returns "" and says why)."""
    import ctypes as C
    from dfdb import _native as N
    lib = N.load()
    buf = C.create_string_buffer(4096)

    def cache_dir():
        assert lib.dfdb_jit_cache_dir(buf, len(buf)) == 0
        return buf.value.decode()

    def why():
        e = C.create_string_buffer(1024)
        lib.dfdb_last_error(e, len(e))
        return e.value.decode()
    monkeypatch.delenv("DFDB_JIT_CACHE", raising=False)
    good = tmp_path / "good"
    monkeypatch.setenv("DFDB_JIT_CACHE_DIR", str(good))
    assert cache_dir() == str(good) and (good.stat().st_mode & 0o777) == 0o700           # created private
    shared = tmp_path / "shared"
    shared.mkdir(); shared.chmod(0o777)
    monkeypatch.setenv("DFDB_JIT_CACHE_DIR", str(shared))
    assert cache_dir() == "" and "writable by group or others" in why()
    shared.chmod(0o770)
    assert cache_dir() == "" and "writable by group or others" in why()
    shared.chmod(0o755)
    assert cache_dir() == str(shared)
    link = tmp_path / "link"
    link.symlink_to(good)
    monkeypatch.setenv("DFDB_JIT_CACHE_DIR", str(link))
    assert cache_dir() == "" and "symbolic link" in why()
    afile = tmp_path / "afile"
    afile.write_bytes(b"x")
    monkeypatch.setenv("DFDB_JIT_CACHE_DIR", str(afile))
    assert cache_dir() == "" and "not a directory" in why()
    if os.geteuid() == 0:                                                              # (only root can make somebody else's directory)
        other = tmp_path / "other"
        other.mkdir(); other.chmod(0o755); os.chown(other, 12345, 12345)
        monkeypatch.setenv("DFDB_JIT_CACHE_DIR", str(other))
        assert cache_dir() == "" and "another user" in why()
    monkeypatch.setenv("DFDB_JIT_CACHE", "0")
    monkeypatch.setenv("DFDB_JIT_CACHE_DIR", str(good))
    assert cache_dir() == "" and "DFDB_JIT_CACHE=0" in why()
