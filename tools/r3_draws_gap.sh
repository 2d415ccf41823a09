cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --no-decode-leg --no-configs 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); k = r['roofline']['kernels']; p = r['config']['placement_calibration']; d = r['default_config']
print('$1  %.4g rows/s  K1 %.4f ms (%.3f)  default K1 %.4f  column cand %.4f-%.4f' % (r['value'], k['scan_cmp']['avg_ms'], r['roofline']['frac'], d['scan_cmp_avg_ms'], p['column_candidates_best_ms'], p['column_candidates_worst_ms']))"; }
for i in 1 2 3 4 5 6 7 8; do one back-to-back; done
for i in 1 2 3 4 5 6 7 8; do sleep 15; one after-15s-idle; done
