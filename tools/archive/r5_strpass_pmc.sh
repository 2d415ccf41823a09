#!/bin/bash
# the String accumulate pass of groupreduce (k_str_pass<2, true>): instruction and stall counters per launch, separate rocprofv3 --pmc passes with --kernel-trace only
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5/strpass_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o h -- python3 $GRAFT_REPO_ROOT/tools/r5_strpass.py 5e8 2 > $OUT/p$i.log 2>&1
done
cd $OUT && python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_str_pass' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
with open('summary.txt', 'w') as o:
    for k in sorted(acc):
        v = acc[k]
        line = "%-42s %-24s n=%2d mean %18.0f  min %18.0f max %18.0f" % (k[0], k[1], len(v), sum(v) / len(v), min(v), max(v))
        print(line); o.write(line + "\n")
PY
