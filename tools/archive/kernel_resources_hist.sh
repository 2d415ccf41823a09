#!/bin/bash
# registers / spills / LDS of K7's history-ring forms alone (a -DDFDB_LZ4_HIST_ONLY build of k_decode.hip: ~1 minute instead of 4)
cd $(dirname $0)/../dataframedbs.jl_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-pass-failed -DDFDB_LZ4_HIST_ONLY --save-temps=obj -c k_decode.hip -o /tmp/kd_hist.o 2>/dev/null
grep -E "^\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size):" /tmp/k_decode-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - - | grep lz4 | sed 's/_ZN4dfdb12k_lz4_decode//; s/ \+/ /g' | cut -c1-200
