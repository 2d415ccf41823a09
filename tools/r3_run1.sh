#!/bin/bash
# round 3, first GPU pass: the new tests, the bench line with every config leg, K2 store-form A/B in the step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_ir_golden.py tests/test_gpu_group.py -m gpu -x -q > gpurun_out/r3/tests_new.log 2>&1; echo "tests_new rc=$?" 
tail -15 gpurun_out/r3/tests_new.log
timeout 900 python bench.py > gpurun_out/r3/bench_full.json 2> gpurun_out/r3/bench_full.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r3/bench_full.err
for cs in 1 3 0 4 1 3; do
  timeout 300 python bench.py --compact-store $cs --no-configs --no-cpu --no-decode-leg --steps 30 >> gpurun_out/r3/k2_ab.jsonl 2>> gpurun_out/r3/k2_ab.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r3/k2_ab.jsonl"):
    r = json.loads(l); k = r["roofline"]["kernels"]
    print("ms/step %.4f  K1 %.4f  K2 %.4f  default: %.4f / K1 %.4f / K2 %.4f" % (r["ms_per_step"], k["scan_cmp"]["avg_ms"], k["compact_indices"]["avg_ms"],
          r["default_config"]["ms_per_step"], r["default_config"]["scan_cmp_avg_ms"], r["default_config"]["compact_indices_avg_ms"]))
PY
