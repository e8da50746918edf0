cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/tmp_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/tmp_stats -- python3 bench.py --main_only --no_cpu_baseline --steps 10 --warmup 3 "$@" > $O/tmp_stats.log 2>&1 || exit 1
python tools/step_kernels.py $O/tmp_stats > $O/tmp_step_kernels.txt
head -16 $O/tmp_step_kernels.txt
