#!/bin/bash
# kernel trace + LDS / wait counters of the 8-bit step (run through gpurun from the repo root).  usage: tools/profile_fp8.sh r03_fp8 [extra bench args]
R=${1:-r03_fp8}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${R}_stats -- python3 bench.py --dtype fp8 --main_only --steps 10 --warmup 3 "$@" > $O/${R}_stats.log 2>&1 || exit 1
python tools/parse_profile.py stats $O/${R}_stats $O/${R}_kernel_stats.csv
python tools/step_kernels.py $O/${R}_stats > $O/${R}_step_kernels.txt
cat $O/${R}_step_kernels.txt
