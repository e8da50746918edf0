#!/bin/bash
# A/B of one cp_config option on ONE box, alternating runs: tools/ab_options.sh finalize_launches [bench args]
# prints ms/step (mean, median) with and without the option, three rounds
opt=$1; shift
for r in 1 2 3; do
  for o in "" "$opt"; do
    CP_BENCH_OPTIONS=$o python bench.py --main_only "$@" 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('options=[$o]', 'ms/step %.4f median %.4f' % (r['ms_per_step'], r['steps_spread']['median_ms']))"
  done
done
