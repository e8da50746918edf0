#!/bin/bash
# A/B of two builds of the library on ONE box, alternating runs: tools/ab_lib.sh build/libcpnative_prev.so [bench args]
# (CPNATIVE_LIB selects the library the package loads; the Python side is the working tree's in both runs)
other=$1; shift
for r in 1 2 3; do
  for l in "" "$other"; do
    CPNATIVE_LIB=${l:+$PWD/$l} python bench.py --main_only "$@" 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('lib=[${l:-working tree}]', 'ms/step %.4f median %.4f' % (r['ms_per_step'], r['steps_spread']['median_ms']))"
  done
done
