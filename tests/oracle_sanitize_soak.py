#!/usr/bin/env python3
"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: GPU sanitizers are not available on this pool): the random selection queues /
projections / aggregates of tests/test_gpu_fuzz.py evaluated by the ORACLE ALONE, plus file round trips through its LZ4 block writer and reader.  The parity
suites trust the oracle's answers; undefined behaviour in it (a signed overflow the compiler may fold, a shift by 64, a read past a block) would make those
answers depend on the compiler.
    python tests/oracle_sanitize_soak.py [--seeds 3000] [--seed0 0]
builds oracle/_san/liboracle.so (gcc -fsanitize=address,undefined -fno-sanitize-recover=undefined) and re-runs itself with the sanitizer runtimes preloaded;
any report ends the run with a non-zero status."""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_DIR = os.path.join(ROOT, "oracle", "_san")


def build():
    os.makedirs(SAN_DIR, exist_ok=True)
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("orc_codec.c", "orc_expr.c", "orc_view.c")]
    out = os.path.join(SAN_DIR, "liboracle.so")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-fPIC", "-D_GNU_SOURCE", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-shared", "-o", out] + srcs + ["-l:liblz4.so.1", "-lm"])
    return out


def child(a):
    for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import numpy as np
    from oracle import oracle as O
    O._LIB_PATH = os.path.join(SAN_DIR, "liboracle.so")
    O.build = lambda force=False: O._LIB_PATH
    from dfdb import ir                      # the IR builder is pure Python
    import test_gpu_fuzz as F                # the generators (its tests need a GPU; the generators do not)

    class OraclePair:
        def __init__(self, cols, block):
            self.names = list(cols); self.nrows = F.N
            self.o = O.Table(block_size=block)
            for k, v in cols.items():
                if isinstance(v, np.ma.MaskedArray):
                    self.o.add_column(k, np.ascontiguousarray(v.filled(0)), missing=np.ma.getmaskarray(v))
                else:
                    self.o.add_column(k, v)

    # the fixture's columns, rebuilt here (tests/test_gpu_fuzz.py: pair)
    N = F.N
    rng = np.random.default_rng(2024)
    f64 = rng.normal(0, 50, N); f64[::101] = np.nan; f64[5::997] = np.inf; f64[7::991] = -0.0
    f32 = rng.normal(0, 8, N).astype(np.float32); f32[::113] = np.nan
    cols = {
        "a": rng.integers(-60, 60, N).astype(np.int64), "b": rng.integers(-2**62, 2**62, N).astype(np.int64), "c": rng.integers(1, 40, N).astype(np.int64),
        "i32": rng.integers(-2**31, 2**31 - 1, N).astype(np.int32), "i8": rng.integers(-128, 127, N).astype(np.int8),
        "u16": rng.integers(0, 2**16 - 1, N).astype(np.uint16), "u64": rng.integers(0, 2**63, N).astype(np.uint64) * np.uint64(2),
        "x": f64, "f": f32, "flag": rng.integers(0, 2, N).astype(bool),
        "m": np.ma.masked_array(rng.integers(-9, 9, N).astype(np.int64), mask=rng.random(N) < 0.25),
        "s": ["%s%d" % ("ab"[i % 2] * (i % 3), i % 23) for i in range(N)],
        "z": rng.integers(-2, 3, N).astype(np.int64),
        "sm": [None if i % 11 == 3 else "%s%d" % ("xy"[i % 2] * (i % 4), i % 7) for i in range(N)],
        "mf": np.ma.masked_array(rng.normal(0, 5, N), mask=rng.random(N) < 0.4),
        "zl": np.where(np.arange(N) >= 2 * N // 3, rng.integers(0, 2, N), rng.integers(1, 5, N)).astype(np.int64),
    }
    pair = OraclePair(cols, F.BLOCK)
    import helpers
    ok = err = refused = 0
    for seed in range(a.seed0, a.seed0 + a.seeds):
        g = F.Gen(ir, seed, risky=seed % 4 == 3)
        stages, proj = g.stages(), g.proj()
        try:
            ov = helpers._oracle_view(pair, stages, proj)
        except Exception:
            refused += 1
            continue
        try:
            ov.nrow(); ov.materialize(); ok += 1
        except Exception:
            err += 1
    # file round trips: the oracle's block writer (liblz4) and reader, ragged block sizes
    with tempfile.TemporaryDirectory() as d:
        for k, bs in enumerate((1, 7, 1000, 65536)):
            t = O.Table(block_size=bs)
            for name in ("a", "x", "s", "sm", "flag", "i8"):
                v = cols[name]
                t.add_column(name, v[:2000] if not isinstance(v, list) else v[:2000])
            t.save(os.path.join(d, "t%d" % k))
            back = O.Table.open(os.path.join(d, "t%d" % k))
            assert back.view().nrow() == 2000
            back.view().materialize()
    print("oracle under ASan + UBSan: %d queues evaluated, %d raised a Julia error, %d refused at build time, 4 file round trips; no sanitizer report" % (ok, err, refused))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3000); ap.add_argument("--seed0", type=int, default=0); ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    build()
    pre = " ".join(subprocess.check_output(["gcc", "-print-file-name=" + n]).decode().strip() for n in ("libasan.so", "libubsan.so"))
    env = dict(os.environ, LD_PRELOAD=pre, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--seeds", str(a.seeds), "--seed0", str(a.seed0)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    sys.stdout.write(r.stdout.decode())
    errtxt = r.stderr.decode()
    if r.returncode != 0 or "runtime error" in errtxt or "AddressSanitizer" in errtxt:
        sys.stderr.write(errtxt[-6000:])
        sys.exit(r.returncode or 1)


if __name__ == "__main__":
    main()
