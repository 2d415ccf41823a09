#!/bin/bash
# K7's history-ring form (tools/r5_hist_sweep.py: compressed-only 1e9-row column, fused decode + predicate): traffic and instruction counters per launch.
# Separate rocprofv3 --pmc passes with --kernel-trace only (the TCC cannot hold FETCH_SIZE and WRITE_SIZE together).  Results under gpurun_out/r5/hist_pmc.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5/hist_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o h -- python3 $GRAFT_REPO_ROOT/tools/r5_hist_sweep.py ${1:-1e9} ${2:-0} > $OUT/p$i.log 2>&1
done
cd $OUT && python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lz4' in r['Kernel_Name']:
            kind = 'hist_scan' if r['Kernel_Name'].rstrip('>').endswith('1') and 'ELi1ELi0ELi6' in r['Kernel_Name'] else r['Kernel_Name'][-60:]
            acc[(r['Kernel_Name'][-48:], r['Counter_Name'])].append(float(r['Counter_Value']))
with open('summary.txt', 'w') as o:
    for k in sorted(acc):
        v = acc[k]
        line = "%-50s %-24s n=%2d mean %18.0f  min %18.0f max %18.0f" % (k[0], k[1], len(v), sum(v) / len(v), min(v), max(v))
        print(line); o.write(line + "\n")
PY
