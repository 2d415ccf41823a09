cd $GRAFT_REPO_ROOT
for c in 8 0 8 0 8 0 8 0 8 0 8 0; do
  python bench.py --no-cpu --no-decode-leg --no-configs --placement-count-candidates $c 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); k = r['roofline']['kernels']; p = r['config']['placement_calibration']
print('count cand $c  %.4g rows/s  K1 %.4f ms (%.3f)  column cand %.4f-%.4f  bitmap cand %.4f-%.4f  count cand %s-%s' % (r['value'], k['scan_cmp']['avg_ms'], r['roofline']['frac'], p['column_candidates_best_ms'], p['column_candidates_worst_ms'], p['candidates_best_ms'], p['candidates_worst_ms'], p.get('count_candidates_best_ms'), p.get('count_candidates_worst_ms')))"
done
