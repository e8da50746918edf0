#!/bin/bash
# A/B of two builds of libcpnative.so on the same box, alternating: tools/ab_bench.sh <other.so> [rounds] [extra bench.py args]
# (the default build against the one given; each leg is one bench.py run without the CPU baseline; the median step time of
#  the timed region is printed next to the mean, because single slow steps move the mean by several per cent)
OTHER=$1
R=${2:-2}
shift; shift
cd "$(dirname "$0")/.."
for i in $(seq $R); do
  for L in default "$OTHER"; do
    if [ "$L" = default ]; then unset CPNATIVE_LIB; else export CPNATIVE_LIB=$PWD/$L; fi
    python bench.py --no_cpu_baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pk=d['roofline']['per_kernel']
sp=d.get('steps_spread', {})
print('%-28s %.3f ms/step (median %.3f, min %.3f)  ' % ('$L', d['ms_per_step'], sp.get('median_ms', 0), sp.get('min_ms', 0)) + '  '.join('%s %.1f' % (k, v['avg_us']) for k, v in pk.items()))"
  done
done
