# ten fresh-process draws of the headline step with and without the placement calibration: K1's average launch and its roofline fraction
for mode in "" "--no-placement"; do
  for i in 1 2 3 4 5 6 7 8 9 10; do [ "$mode" != "" ] && [ $i -gt 3 ] && continue;
    python bench.py --steps 20 --warmup 3 --no-cpu --no-decode-leg $mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('$mode'.ljust(15), 'k1_ms %.4f frac %.3f ms_per_step %.4f' % (r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['ms_per_step']), r['config']['placement_calibration'] if isinstance(r['config']['placement_calibration'], str) else {k: round(v,4) for k,v in r['config']['placement_calibration'].items() if isinstance(v,float)})
"
  done
done
