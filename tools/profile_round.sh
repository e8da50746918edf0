#!/bin/bash
# One profiling round on the GPU box (run through gpurun from the repo root): kernel stats, HBM traffic (FETCH_SIZE and
# WRITE_SIZE in separate passes), MFMA busy counters (their own pass), behind the plain bench line.  usage: tools/profile_round.sh r03 [bench args]
# Every rocprofv3 call has the program directly behind "--" (no env/bash hop) and --pmc passes carry --kernel-trace only.
R=${1:-r02}; shift     # further arguments go to every bench.py run (e.g. --dtype fp8)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
# the plain bench line first: behind the four profiled runs the chip is warm and the same build reads 2-4 % slower
timeout -k 10 400 python bench.py "$@" > $O/${R}_bench.log 2>&1; tail -1 $O/${R}_bench.log > $O/${R}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${R}_stats -- python3 bench.py --main_only --steps 10 --warmup 3 "$@" > $O/${R}_stats.log 2>&1 || exit 1
# counter passes (one stream, the default: a kernel's HBM bytes and matrix-pipe cycles are read over its own execution)
for Cn in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 280 rocprofv3 --pmc $Cn --kernel-trace -d $O/${R}_$Cn -- python3 bench.py --main_only --steps 4 --warmup 2 "$@" > $O/${R}_$Cn.log 2>&1 || exit 1
done
timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/${R}_SQ -- python3 bench.py --main_only --steps 4 --warmup 2 "$@" > $O/${R}_SQ.log 2>&1 || echo "SQ pass failed (see $O/${R}_SQ.log)"
python tools/parse_profile.py stats $O/${R}_stats $O/${R}_kernel_stats.csv
python tools/step_kernels.py $O/${R}_stats > $O/${R}_step_kernels.txt
python tools/step_kernels.py $O/${R}_stats --timeline > $O/${R}_step_timeline.txt
python tools/step_kernels.py $O/${R}_stats --by_position gemm_tn > $O/${R}_wgrad_by_position.txt
python tools/parse_profile.py traffic $O/${R}_FETCH_SIZE $O/${R}_WRITE_SIZE $O/${R}_traffic.json
python tools/parse_profile.py mfma $O/${R}_SQ $O/${R}_mfma.json || true
# the raw traces stay on the box: gpurun merges at most 64 MiB back
rm -rf $O/${R}_stats $O/${R}_FETCH_SIZE $O/${R}_WRITE_SIZE $O/${R}_SQ
