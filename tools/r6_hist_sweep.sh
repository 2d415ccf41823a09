#!/bin/bash
# K7's history-ring scan (compressed-only 1e9-row Int64 column, fused decode + predicate): ms per launch over the number of rings and the kernel shape
# (DFDB_LZ4_HIST_VARIANT), then FETCH_SIZE / WRITE_SIZE per launch for each number of rings (separate --pmc passes, kernel trace only).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6/hist
mkdir -p $OUT
WAVES=2048,3072,4096,0
for v in 0 1 2; do
  DFDB_LZ4_HIST_VARIANT=$v timeout 600 python3 $GRAFT_REPO_ROOT/tools/hist_sweep.py 1e9 $WAVES > $OUT/ms_variant$v.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o h -- python3 $GRAFT_REPO_ROOT/tools/hist_sweep.py 1e9 $WAVES > $OUT/pmc_$c.log 2>&1
done
cd $OUT && python3 - <<'PY' > $OUT/pmc_summary.txt
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob('pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'lz4_decode' in r['Kernel_Name'] and r['Counter_Name'] == c:
                rows.append((int(r.get('Dispatch_Id', 0) or 0), int(r.get('Grid_Size', 0) or 0), float(r['Counter_Value'])))
    rows.sort()
    for d, g, v in rows: print(c, 'dispatch', d, 'grid', g, 'value', v)
PY
