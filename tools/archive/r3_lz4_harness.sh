#!/bin/bash
# round 3: K7 after the superbatch change — the harness table of profiles/r3_lz4_harness.txt, then the PMC passes of profiles/r3_pmc_lz4.txt
# (tools/bench_lz4_noprof: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idataframedbs.jl_amd/csrc tools/bench_lz4.hip -o tools/bench_lz4_noprof -ldl)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
{
for args in "15259 0 -1" "15259 0 10" "1526 0 -1" "1526 0 0" "2048 0 -1" "2560 0 -1"; do echo "== bench_lz4_noprof $args"; tools/bench_lz4_noprof $args | tail -3; done
for m in 1 2 5 6 3 4; do echo "== bench_lz4_noprof 8192 $m -1"; tools/bench_lz4_noprof 8192 $m -1 | tail -1; echo "== bench_lz4_noprof 8192 $m 10 (round 2 shape)"; tools/bench_lz4_noprof 8192 $m 10 | tail -1; done
} > gpurun_out/r3/lz4_harness.txt 2>&1
cat gpurun_out/r3/lz4_harness.txt | grep -v "^blocks" | paste - - | head -40
bash tools/r3_k7_pmc.sh 15259 -1 > /dev/null 2>&1; cat gpurun_out/r3/k7pmc_15259/summary.txt | grep -E "INSTS_(VALU|SALU|LDS) |WAVE_CYCLES|RDREQ|WRREQ|LDS_IDX|BANK"
