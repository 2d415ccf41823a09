#!/bin/bash
# VGPRs / spills / LDS of every kernel in one .hip source, compiled for gfx950 (device code only): tools/kernel_resources.sh k_scan.hip [name filter]
set -e
SRC=$1; FILTER=${2:-.}
D=$(dirname "$0")/../dataframedbs.jl_amd/csrc
OUT=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-pass-failed --cuda-device-only -c "$D/$SRC" -o "$OUT/k.co"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$OUT/k.co" --output="$OUT/k.elf" --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$OUT/k.elf" | grep -E "\.name:|\.vgpr_count|vgpr_spill|group_segment_fixed|\.sgpr_count" | paste - - - - - | grep -E "$FILTER" | sed -E 's/ +/ /g'
rm -rf "$OUT"
