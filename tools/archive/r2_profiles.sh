# Round-2 profile collection on one MI355X (run through gpurun from the repo root; everything lands in gpurun_out/r2p/).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel stats of the headline bench + the line that process printed
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
# 2. the default line (cpu_baseline + decode-inclusive leg)
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
# 3. HBM traffic of K1 / K2 / K7: separate PMC passes (the TCC cannot hold FETCH_SIZE and WRITE_SIZE together)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
done
# 4. the functional N-rank lines on this 1-GPU box and the C-ABI exchange
python3 $R/bench.py --gpus 2 --all-on-device0 --backend gloo --rows 200000000 --steps 5 --warmup 1 --no-cpu > $O/bench_2ranks_gloo_device0.json 2>/dev/null
python3 $R/bench.py --exchange lib --steps 10 --warmup 2 --no-cpu --no-decode-leg > $O/bench_exchange_lib.json 2>/dev/null
# 5. LZ4 harness + instruction counters per sequence
( for n in 15259 1526; do $R/tools/bench_lz4_noprof $n 0; done; for pm in 0 1; do echo pipe=$pm; $R/tools/bench_lz4_noprof 1526 0 $pm; $R/tools/bench_lz4_noprof 15259 0 $pm; done; for m in 1 2 5 6 3 4; do $R/tools/bench_lz4_noprof 8192 $m; done; $R/tools/bench_lz4 1526 0 ) > $O/lz4_harness.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_lz4 -o k7 -- $R/tools/bench_lz4_noprof 1526 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_lz4_1w -o k7 -- $R/tools/bench_lz4_noprof 15259 0 > /dev/null 2>&1
# 6. configs 2-4, interpreter, LZ4 through the engine, under per-kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg -o c -- python3 $R/tools/bench_configs.py --reps 3 --lz4-rows 100000000 > $O/configs_bench.jsonl 2> $O/configs_bench.err
python3 $R/tools/bench_config5.py > $O/config5_one_gpu.json 2> $O/config5.err
ls -R $O | head -50
