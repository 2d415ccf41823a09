#!/bin/bash
# What IS a TCC_EA0_RDREQ on gfx950?  The 32- / 64- / 128-byte breakdown of the memory-side read requests of K7 (sparse 24-byte far reads + a compressed stream)
# and of the bench step (K1: a pure stream; K2), plus the DRAM-destined share.  Separate passes, kernel trace only.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3/reqsize; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_READ_SECTORS_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/k7_$i -o p -- $GRAFT_REPO_ROOT/tools/bench_lz4_noprof 15259 0 -1 > $OUT/k7_$i.log 2>&1
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/step_$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-configs --no-decode-leg > $OUT/step_$i.log 2>&1
done
cd $OUT && python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("k_lz4_decode", "k_scan_cmp", "k_compact_indices"):
            if k in r["Kernel_Name"]:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("summary.txt", "w") as o:
    for (k, c), v in sorted(acc.items()):
        v = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v
        line = "%-20s %-26s %16.0f per launch (%d launches)" % (k, c, sum(v) / len(v), len(v))
        print(line); o.write(line + "\n")
PY
