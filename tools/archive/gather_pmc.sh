cd /tmp && export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/bench_gather
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE"; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gpmc -o g -- $GRAFT_REPO_ROOT/tools/bench_gather > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/gpmc'
for f in glob.glob(root+'/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_gather' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k in sorted(acc): print(k, sum(acc[k])/len(acc[k]))
PY
done
