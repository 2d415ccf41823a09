#!/bin/bash
# K7 counters on the standalone harness (tools/bench_lz4_noprof NBLOCKS MODE PIPE): separate --pmc passes, kernel trace only
NB=${1:-15259}; PIPE=${2:--1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3/k7pmc_$NB
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o k7 -- $GRAFT_REPO_ROOT/tools/bench_lz4_noprof $NB 0 $PIPE > $OUT/p$i.log 2>&1
done
cd $OUT && python3 - <<'PY'
import csv, glob, collections, os
nb = int(os.path.basename(os.getcwd()).split('_')[-1])
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lz4' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
seqs = nb * 65527
with open('summary.txt', 'w') as o:
    for k in sorted(acc):
        line = "%-28s %16.0f per dispatch = %10.3f per sequence" % (k, acc[k] / n[k], acc[k] / n[k] / seqs)
        print(line); o.write(line + "\n")
PY
