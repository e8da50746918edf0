#!/bin/bash
# bench.py with the static and the dynamic tile schedule, alternating, on the same box: tools/ab_sched.sh [rounds]
R=${1:-2}
cd "$(dirname "$0")/.."
for i in $(seq $R); do
  for S in static dynamic; do
    CPNATIVE_TILE_SCHEDULE=$S python bench.py --no_cpu_baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pk=d['roofline']['per_kernel']
print('%-8s %.3f ms/step  ' % (d['config']['tile_schedule'], d['ms_per_step']) + '  '.join('%s %.1f' % (k, v['avg_us']) for k, v in pk.items()))"
  done
done
