# Round-4 profile collection on one MI355X (run through gpurun from the repo root; everything lands in gpurun_out/r4p/).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel stats of the bench command (every leg) + the line that process printed
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
# 1b. nothing but the timed step in the process: the trace's k_scan_cmp average and the line's roofline.avg_launch_ms are the same launches
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step -o b -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-configs --no-decode-leg --no-cold > $O/bench_step_under_rocprof.json 2> $O/bench_step_under_rocprof.err
# 2. HBM traffic of K1 / K2 / K7: separate PMC passes (the TCC cannot hold FETCH_SIZE and WRITE_SIZE together)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-configs --no-cold > /dev/null 2>&1
done
# 3. the interpreter against its run-time compiled kernels: instructions per launch
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_interp -o p -- python3 $R/tools/r4_interp.py --reps 2 > $O/interp_under_pmc.json 2> $O/interp_under_pmc.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_interp_$c -o p -- python3 $R/tools/r4_interp.py --reps 2 > /dev/null 2>&1
done
python3 $R/tools/r4_interp.py > $O/interp.json 2> $O/interp.err
# 4. narrow column types
python3 $R/tools/diag_types.py 1e9 > $O/types.txt 2>&1
# 5. functional N-rank lines on this 1-GPU box: eight self-spawned ranks, and one process driving three shards
python3 $R/bench.py --gpus 8 --all-on-device0 --backend gloo --rows 50000000 --steps 5 --warmup 1 --no-cpu --config-scale 0.01 > $O/bench_8ranks_gloo_device0.json 2> $O/bench_8ranks.err
python3 $R/bench.py --mode threads --gpus 3 --all-on-device0 --rows 200000000 --steps 5 --warmup 1 --config-scale 0.05 > $O/bench_threads3_device0.json 2> $O/bench_threads3.err
ls -R $O | head -60
