#!/bin/bash
# A/B of one environment switch of the PYTHON side on ONE box, alternating runs: tools/ab_envvar.sh CPNATIVE_AUX_STREAM=0 [bench args]
kv=$1; shift
for r in 1 2 3; do
  for e in "" "$kv"; do
    env $e python bench.py --main_only "$@" 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('env=[$e]', 'ms/step %.4f median %.4f loss %.5f' % (r['ms_per_step'], r['steps_spread']['median_ms'], r['loss']))"
  done
done
