#!/bin/bash
# K7 with the sequence-start index (tools/bench_lz4_noprof NBLOCKS MODE PIPE 1): per-sequence instruction and busy counters of the RECORDING launch and of the INDEXED ones
NB=${1:-15259}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3/k7idx_$NB
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INSTS_BRANCH" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o k7 -- $GRAFT_REPO_ROOT/tools/bench_lz4_noprof $NB 0 0 1 > $OUT/p$i.log 2>&1
done
cd $OUT && python3 - <<'PY'
import csv, glob, collections, os
nb = int(os.path.basename(os.getcwd()).split('_')[-1])
acc = collections.defaultdict(list)
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'lz4' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    for r in rows:
        which = 'recording' if int(r['Dispatch_Id']) == ids[0] else 'indexed'
        acc[(which, r['Counter_Name'])].append(float(r['Counter_Value']))
seqs = nb * 65527
with open('summary.txt', 'w') as o:
    for k in sorted(acc):
        v = acc[k]
        line = "%-10s %-24s %16.0f per dispatch = %10.3f per sequence" % (k[0], k[1], sum(v) / len(v), sum(v) / len(v) / seqs)
        print(line); o.write(line + "\n")
PY
