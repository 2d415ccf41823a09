#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python tools/r3_k2_ab.py >> gpurun_out/r3/k2_ab_inprocess.jsonl 2>> gpurun_out/r3/k2_ab_inprocess.err; done
python - <<'PY'
import json
for l in open("gpurun_out/r3/k2_ab_inprocess.jsonl"):
    r = json.loads(l)
    print({k: (v["k2"]["median"], v["k1"]["median"], v["step"]["median"]) for k, v in r["forms"].items()})
PY
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3/tests_all.log 2>&1; echo "tests_all rc=$?"
tail -25 gpurun_out/r3/tests_all.log
