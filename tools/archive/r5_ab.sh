#!/bin/bash
# same box, alternating: the library in the tree (A) against the one under _xp/ (B) — kernel times by rocprofv3 --kernel-trace, the script's own JSON lines beside them
# usage: tools/r5_ab.sh [script (tools/r5_strpass.py)] [kernel substring (k_str_pass<2)] [script args ...]
SCRIPT=${1:-tools/r5_strpass.py}; KERNEL=${2:-k_str_pass<2}; shift 2 2>/dev/null
ARGS=${@:-5e8 3}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5; cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  for v in A B; do
    if [ $v = B ]; then export DFDB_PKG=_xp; else unset DFDB_PKG; fi
    rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5/ab_$v$i -o s -- python3 $GRAFT_REPO_ROOT/$SCRIPT $ARGS > $GRAFT_REPO_ROOT/gpurun_out/r5/ab_$v$i.log 2>&1
  done
done
KERNEL="$KERNEL" python3 - <<'PY'
import csv, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5/"
for v in "AB":
    for i in (1, 2, 3):
        t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(root + f"ab_{v}{i}/s_kernel_trace.csv")) if os.environ["KERNEL"] in r["Kernel_Name"]]
        wall = [l.strip()[:70] for l in open(root + f"ab_{v}{i}.log") if l.startswith("{")]
        print(v, i, "kernel min %.3f ms (n=%d)" % (min(t) if t else -1, len(t)), *wall)
PY
