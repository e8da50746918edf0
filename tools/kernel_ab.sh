#!/bin/bash
# one traced step per library, selected kernels: tools/kernel_ab.sh 'regex' lib1.so lib2.so ...   (default library first)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pat=$1; shift
for L in default "$@"; do
  if [ "$L" = default ]; then unset CPNATIVE_LIB; else export CPNATIVE_LIB=$PWD/$L; fi
  rm -rf gpurun_out/kab
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kab -- python3 bench.py --main_only --no_cpu_baseline --steps 10 --warmup 3 $BENCH_ARGS > gpurun_out/kab.log 2>&1 || exit 1
  echo "== $L"; python tools/step_kernels.py gpurun_out/kab | grep -E "step span|$pat"
done
rm -rf gpurun_out/kab
