#!/bin/bash
# PMC counters of the three radix passes of unique (k_radix.hip) at 1e9 rows / 1e6 distinct values: separate --pmc passes, kernel trace only.
# $1 = "mem" collects only the HBM traffic counters (one counter per pass, as the guide's HBM section prescribes); $2 = "group": groupreduce over 5e4 groups instead of unique
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6/radix_pmc
rm -rf $OUT; mkdir -p $OUT
if [ "$1" = "mem" ]; then SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum")
else SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"); fi
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  PROG=$GRAFT_REPO_ROOT/tools/r6_radix_xp.py; if [ "$2" = "group" ]; then PROG=$GRAFT_REPO_ROOT/tools/r6_groupreduce_many.py; export DFDB_ONE_LEG=1; fi
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o r -- python3 $PROG > $OUT/p$i.log 2>&1
  echo "pass $i ($set): exit $?" >> $OUT/passes.txt
done
cd $OUT && python3 - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r6/radix_pmc.txt
import csv,glob,collections,re
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r'k_radix_\w+', r['Kernel_Name'])
        if m:
            acc[(m.group(0),r['Counter_Name'])]+=float(r['Counter_Value']); n[(m.group(0),r['Counter_Name'])]+=1
for k in sorted(acc): print('%-20s %-24s %.4g per dispatch (%d dispatches)' % (k[0], k[1], acc[k]/n[k], n[k]))
print(open('passes.txt').read())
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r6/radix_pmc.txt
