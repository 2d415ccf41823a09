#!/bin/bash
# The first run on a node with more than one MI355X: nothing in this repository has met two physical GPUs yet.
#   1. RCCL smoke: every device in one dfdb_group_create, one all-reduce per operator of the rank ids
#   2. the multi-device GPU tests (tests/test_gpu_multidevice.py: real RCCL between distinct devices, answers == oracle)
#   3. bench.py at N = 1, 2, 4, 8 (as many as there are) in three forms — one process per GPU with torch.distributed's nccl, the same with the library's own
#      communicator (--exchange lib), and ONE process driving every GPU (--mode threads) — one JSON file per N and form under $OUT
# Every program is started as a plain child process; nothing re-executes itself after touching a GPU.  To profile a form, put the program straight after
# `rocprofv3 ... --`, e.g.   rocprofv3 --kernel-trace --stats -d $OUT/prof -- python3 bench.py --mode threads --gpus 8
# (one process: the N-rank forms spawn ranks and are profiled per rank by the launcher of your choice, never through `env` / `bash -c`).
set -u
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/multigpu}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python3 -c 'import torch; print(torch.cuda.device_count())')
echo "devices: $NDEV" | tee "$OUT/devices.txt"
if [ "$NDEV" -lt 2 ]; then echo "one device: nothing to do here (the single-GPU suite is tests/ -m gpu)"; exit 0; fi
echo "== 1. RCCL smoke + 2. multi-device tests"
python3 -m pytest tests/test_gpu_multidevice.py -x -q -m gpu 2>&1 | tee "$OUT/pytest_multidevice.log" | tail -15
rc_tests=${PIPESTATUS[0]}
echo "== 3. bench.py"
for n in 1 2 4 8; do
  [ "$n" -gt "$NDEV" ] && break
  python3 bench.py --gpus $n --steps 20 --warmup 5 > "$OUT/bench_torch_n$n.json" 2> "$OUT/bench_torch_n$n.err"; echo "torch nccl  N=$n rc=$? $(python3 tools/show_bench.py "$OUT/bench_torch_n$n.json" 2>/dev/null | head -1)"
  if [ "$n" -gt 1 ]; then
    python3 bench.py --gpus $n --exchange lib --steps 20 --warmup 5 --no-cpu > "$OUT/bench_lib_n$n.json" 2> "$OUT/bench_lib_n$n.err"; echo "library RCCL N=$n rc=$? $(python3 tools/show_bench.py "$OUT/bench_lib_n$n.json" 2>/dev/null | head -1)"
    python3 bench.py --mode threads --gpus $n --steps 20 --warmup 5 > "$OUT/bench_threads_n$n.json" 2> "$OUT/bench_threads_n$n.err"; echo "one process  N=$n rc=$? $(python3 tools/show_bench.py "$OUT/bench_threads_n$n.json" 2>/dev/null | head -1)"
  fi
done
python3 - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, "bench_*_n*.json"))):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][0])
        rows.append((os.path.basename(f), r["n_gpus"], r["value"], r["ms_per_step"], (r.get("roofline") or {}).get("frac")))
    except Exception as e:
        rows.append((os.path.basename(f), None, None, None, str(e)))
base = {}
for name, n, v, ms, fr in rows:
    form = name.split("_n")[0]
    if n == 1: base[form] = v
print("%-24s %3s %14s %9s %7s %s" % ("file", "N", "rows/s", "ms/step", "K1 frac", "x vs N=1 (torch form's N=1 for the others)"))
b1 = base.get("bench_torch")
for name, n, v, ms, fr in rows:
    if v is None: print(name, "FAILED:", fr); continue
    print("%-24s %3d %14.4g %9.3f %7.3f %s" % (name, n, v, ms, fr or 0, ("%.2f" % (v / b1)) if b1 else ""))
PY
exit $rc_tests
