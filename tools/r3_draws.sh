#!/bin/bash
# ten fresh-process draws of the headline step on one box: calibrated (the line's `value`) and the library's default options (`default_config`)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
for i in 1 2 3 4 5 6 7 8 9 10; do
  python bench.py --no-cpu --no-decode-leg --no-configs 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); k = r['roofline']['kernels']; d = r['default_config']
print('draw $i  calibrated: %.4g rows/s  step %.4f ms  K1 %.4f ms (%.3f)  K2 %.4f ms  job %.0f GB/s | default options: %.4g rows/s  step %.4f ms  K1 %.4f ms (%.3f)' % (
    r['value'], r['ms_per_step'], k['scan_cmp']['avg_ms'], r['roofline']['frac'], k['compact_indices']['avg_ms'], r['job_hbm_gbps'], d['value'], d['ms_per_step'], d['scan_cmp_avg_ms'], d['roofline_frac']))"
done | tee gpurun_out/r3/placement_draws.txt
