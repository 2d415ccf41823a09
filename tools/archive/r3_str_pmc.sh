#!/bin/bash
# counters of the flat string scan (k_str_match_short over 5e8 rows) beside K1's on the same box: where do its 4.6 TB/s come from?  Separate --pmc passes, kernel trace only.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3/strpmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o s -- python3 $GRAFT_REPO_ROOT/tools/r3_str.py --reps 3 > $OUT/p$i.log 2>&1
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/k$i -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-configs --no-decode-leg > $OUT/k$i.log 2>&1
done
cd $OUT && python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("k_str_match_short", "k_scan_cmp"):
            if k in r["Kernel_Name"]:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("summary.txt", "w") as o:
    for (k, c), v in sorted(acc.items()):
        v = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v
        line = "%-20s %-40s %18.0f per launch (%d launches)" % (k, c, sum(v) / len(v), len(v))
        print(line); o.write(line + "\n")
PY
