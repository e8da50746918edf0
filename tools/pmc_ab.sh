#!/bin/bash
# HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes) of two builds of libcpnative.so on one box:
#   gpurun -- 'bash tools/pmc_ab.sh path/to/other/libcpnative.so'   ->  gpurun_out/traffic_{default,other}.json
OTHER=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for V in default other; do
  if [ $V = other ]; then export CPNATIVE_LIB=$GRAFT_REPO_ROOT/$OTHER; fi
  for Cn in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 280 rocprofv3 --pmc $Cn --kernel-trace -d gpurun_out/pmc_${V}_$Cn -- python3 bench.py --no_cpu_baseline --steps 4 --warmup 2 > gpurun_out/pmc_${V}_$Cn.log 2>&1 || exit 1
  done
  python tools/parse_profile.py traffic gpurun_out/pmc_${V}_FETCH_SIZE gpurun_out/pmc_${V}_WRITE_SIZE gpurun_out/traffic_$V.json
done
