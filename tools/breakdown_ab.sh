#!/bin/bash
# per-kernel-kind step breakdown (bench.py --breakdown) of the default library and of others: tools/breakdown_ab.sh lib1.so lib2.so ... [-- kinds-regex]
cd "$(dirname "$0")/.."
for L in default "$@"; do
  if [ "$L" = default ]; then unset CPNATIVE_LIB; else export CPNATIVE_LIB=$PWD/$L; fi
  echo "== $L"
  python bench.py --no_cpu_baseline --breakdown 2>&1 >/dev/null | grep -E "${KINDS:-conv|sum}"
done
