#!/bin/bash
# same box, alternating: the library in the tree (A) against the one under _xp/ (B) on tools/r5_strpass.py — kernel times of k_str_pass<2, ...> by rocprofv3 --kernel-trace
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5; cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  for v in A B; do
    if [ $v = B ]; then export DFDB_PKG=_xp; else unset DFDB_PKG; fi
    rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5/ab_$v$i -o s -- python3 $GRAFT_REPO_ROOT/tools/r5_strpass.py ${1:-5e8} 3 > $GRAFT_REPO_ROOT/gpurun_out/r5/ab_$v$i.log 2>&1
  done
done
python3 - <<'PY'
import csv, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5/"
for v in "AB":
    for i in (1, 2, 3):
        t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(root + f"ab_{v}{i}/s_kernel_trace.csv")) if "k_str_pass<2" in r["Kernel_Name"]]
        wall = [l for l in open(root + f"ab_{v}{i}.log") if l.startswith("{")]
        print(v, i, "accumulate kernel min %.3f ms" % min(t), wall[-1].strip()[:60] if wall else "")
PY
