#!/bin/bash
# PMC counters of the three radix passes of unique (k_radix.hip) at 1e9 rows / 1e6 distinct values: separate --pmc passes, kernel trace only
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6/radix_pmc
mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE WRITE_SIZE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o r -- python3 $GRAFT_REPO_ROOT/tools/r6_radix_xp.py > $OUT/p$i.log 2>&1
done
cd $OUT && python3 - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r6/radix_pmc.txt
import csv,glob,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'radix' in k:
            name=k.split('(')[0].split('::')[-1][:24]
            acc[(name,r['Counter_Name'])]+=float(r['Counter_Value']); n[(name,r['Counter_Name'])]+=1
for k in sorted(acc): print(k[0], k[1], acc[k]/n[k], 'per dispatch (%d dispatches)' % n[k])
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r6/radix_pmc.txt
