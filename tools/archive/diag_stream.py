"""Per-chunk timing of the streaming loader (DFDB_STREAM_DEBUG=1 python tools/diag_stream.py [rows] [chunk_blocks])."""
import os, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dataframedbs.jl_amd")):
    sys.path.insert(0, p)
import torch, dfdb  # noqa
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(2.5e8)
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = tempfile.mkdtemp(dir="/dev/shm")
try:
    t = dfdb.DFTable.new(); t.add_generated("x", dfdb.GEN_I64_MOD1M, 1, n); t.save(d + "/tb"); t.close()
    tb = dfdb.open_table(d + "/tb", load=False)
    v = tb[("x", lambda x: x > 899999), dfdb.ALL]
    for rep in range(2):
        t0 = time.perf_counter()
        s = dfdb.stream(v, chunk)
        t1 = time.perf_counter()
        tot = 0
        with s:
            for part in s:
                tot += part.count()
        t2 = time.perf_counter()
        print("rep", rep, "rows", tot, "open %.1f ms" % ((t1 - t0) * 1e3), "iterate %.1f ms" % ((t2 - t1) * 1e3), file=sys.stderr)
finally:
    shutil.rmtree(d)
