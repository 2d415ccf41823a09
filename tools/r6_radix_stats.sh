#!/bin/bash
# rocprofv3 --kernel-trace --stats of unique by radix at 1e9 rows / 1e6 distinct values (tools/r6_radix_xp.py): per-kernel averages -> gpurun_out/r6/radix_stats.csv
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6/radix_stats; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python3 $GRAFT_REPO_ROOT/tools/r6_radix_xp.py > $O/run.log 2>&1
f=$(find $O -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r6/radix_stats.csv
grep -i "radix\|unique\|Name" $GRAFT_REPO_ROOT/gpurun_out/r6/radix_stats.csv | cut -c1-220
grep "^XP" $O/run.log
