# What the bitmap placement calibration costs and buys, by spacer size and candidate count: fresh process per line (run through gpurun).
cd $GRAFT_REPO_ROOT
for cfg in "--no-placement" "--placement-spacer-mb 0" "--placement-spacer-mb 256" "--placement-spacer-mb 2048" "--placement-spacer-mb 12288" "--placement-spacer-mb 2048 --placement-candidates 4" "--placement-spacer-mb 12288 --placement-candidates 4"; do
  for rep in 1 2 3; do
    python3 bench.py --steps 10 --warmup 2 --no-cpu --no-decode-leg $cfg 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['config']['placement_calibration']
        print('$cfg'.ljust(60), 'K1 %.4f ms  frac %.3f ' % (d['roofline']['avg_launch_ms'], d['roofline']['frac']), p if p == 'off' else 'one-time %.2f s  best %.4f worst %.4f' % (p['one_time_seconds'], p['candidates_best_ms'], p['candidates_worst_ms']))
"
  done
done
