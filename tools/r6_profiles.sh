# Round-6 profile collection on one MI355X (run through gpurun from the repo root; everything lands in gpurun_out/r6p/).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel stats of the bench command (every leg) + the line that process printed
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
# 1b. nothing but the timed step in the process: the trace's k_scan_cmp average and the line's roofline.avg_launch_ms are the same launches
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step -o b -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-configs --no-decode-leg --no-cold > $O/bench_step_under_rocprof.json 2> $O/bench_step_under_rocprof.err
# 2. HBM traffic of K1 / K2 / K7 (incl. the history-ring form of the arena leg): separate PMC passes (the TCC cannot hold FETCH_SIZE and WRITE_SIZE together)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-configs --no-cold > /dev/null 2>&1
done
ls -R $O | head -40
