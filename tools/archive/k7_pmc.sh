cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/k7pmc -o k7 -- $GRAFT_REPO_ROOT/tools/bench_lz4 1526 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT/gpurun_out/k7pmc && python3 - <<'PY'
import csv,glob,collections
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if 'lz4' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
    for k in acc: print(k, acc[k]/n[k], 'per dispatch; per seq', acc[k]/n[k]/(1526*65527))
PY
