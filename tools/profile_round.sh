cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01f_stats -- python3 bench.py --no_cpu_baseline --steps 10 --warmup 3 > gpurun_out/r01f_stats.log 2>&1 || exit 1
for Cn in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 280 rocprofv3 --pmc $Cn --kernel-trace -d gpurun_out/r01f_$Cn -- python3 bench.py --no_cpu_baseline --steps 4 --warmup 2 > gpurun_out/r01f_$Cn.log 2>&1 || exit 1
done
python tools/parse_profile.py stats gpurun_out/r01f_stats gpurun_out/r01f_kernel_stats.csv
python tools/parse_profile.py traffic gpurun_out/r01f_FETCH_SIZE gpurun_out/r01f_WRITE_SIZE gpurun_out/r01f_traffic.json
timeout -k 10 400 python bench.py > gpurun_out/r01f_bench.log 2>&1; tail -1 gpurun_out/r01f_bench.log > gpurun_out/r01f_bench.json
