#!/bin/bash
# instruction / stall / LDS / cache counters of one kernel, separate rocprofv3 --pmc passes with --kernel-trace only
# usage: tools/r5_pmc.sh <script> <kernel substring> <out name> [script args ...]
SCRIPT=$1; KERNEL=$2; NAME=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5/pmc_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS" "FETCH_SIZE" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o h -- python3 $GRAFT_REPO_ROOT/$SCRIPT "$@" > $OUT/p$i.log 2>&1
done
cd $OUT && KERNEL="$KERNEL" python3 - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(list)
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ["KERNEL"] in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
with open('summary.txt', 'w') as o:
    for k in sorted(acc):
        v = acc[k]
        line = "%-50s %-30s n=%2d mean %18.0f  min %18.0f max %18.0f" % (k[0], k[1], len(v), sum(v) / len(v), min(v), max(v))
        print(line); o.write(line + "\n")
PY
