# Round-3 profile collection on one MI355X (run through gpurun from the repo root; everything lands in gpurun_out/r3p/).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel stats of the bench command (config legs included) + the line that process printed
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
# 1b. the same with nothing but the timed step in the process (no calibration, no config / decode legs): every k_scan_cmp launch is a warm-up or a timed step,
#     so the trace's average and the line's roofline.avg_launch_ms are the same launches
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step -o b -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-configs --no-decode-leg > $O/bench_step_under_rocprof.json 2> $O/bench_step_under_rocprof.err
# 2. the default line (what the driver runs)
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
# 3. HBM traffic of K1 / K2 / K7: separate PMC passes (the TCC cannot hold FETCH_SIZE and WRITE_SIZE together)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-configs > /dev/null 2>&1
done
# 4. the functional N-rank lines on this 1-GPU box and the C-ABI exchange
python3 $R/bench.py --gpus 2 --all-on-device0 --backend gloo --rows 200000000 --steps 5 --warmup 1 --no-cpu --config-scale 0.05 > $O/bench_2ranks_gloo_device0.json 2> $O/bench_2ranks.err
python3 $R/bench.py --exchange lib --steps 10 --warmup 2 --no-cpu --no-decode-leg > $O/bench_exchange_lib.json 2> $O/bench_exchange_lib.err
python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-decode-leg --config5-host-shards 4 > $O/bench_config5_host_shards4.json 2> $O/bench_host4.err
ls -R $O | head -40
