#!/bin/bash
# one rocprofv3 --pmc pass over a short bench run (through gpurun, from the repo root).
# usage: tools/pmc_pass.sh <tag> "<counters>" [bench args]      ->  gpurun_out/<tag>.json
R=$1; CN=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 280 rocprofv3 --pmc $CN --kernel-trace -d $O/${R}_pmc -- python3 bench.py --main_only --steps 3 --warmup 2 "$@" > $O/${R}_pmc.log 2>&1 || { tail -5 $O/${R}_pmc.log; exit 1; }
python tools/parse_profile.py counters $O/${R}_pmc $O/${R}.json
