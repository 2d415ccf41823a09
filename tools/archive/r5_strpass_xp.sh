cd /tmp && export TMPDIR=/tmp
for x in 0 12 14 40 41 43 47 24 56 59 63; do
  export DFDB_XP_STR=$x
  DFDB_PKG=_xp rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5/xp_$x -o s -- python3 $GRAFT_REPO_ROOT/tools/r5_strpass.py 5e8 3 > /dev/null 2>&1
done
