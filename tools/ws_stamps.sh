#!/bin/bash
# The power-cap evidence of DESIGN.md section 4 ("Weight-stationary GEMMs"): the 32x32x16 weight-stationary forward kernel with
# in-kernel stamps (s_memtime = shader cycles, s_memrealtime = 100 MHz) in four builds: everything, without the epilogue, without the
# row fetches, without both (MFMAs + fragment reads alone).
#   tools/ws_stamps.sh build     (build host: four tools-only libraries into build/)
#   tools/ws_stamps.sh run       (GPU box, through gpurun: prints the table -> profiles/rNN_ws_stamps.txt)
cd "$(dirname "$0")/.."
V=("all:" "no_epi:-DWS_NO_EPI" "no_fetch:-DWS_NO_FETCH" "mfma_only:-DWS_NO_EPI -DWS_NO_FETCH")
if [ "$1" = build ]; then
  mkdir -p build
  for v in "${V[@]}"; do
    n=${v%%:*}; f=${v#*:}
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCP_VARIANTS -DWS_STAMP $f contrastiveprosthetics_amd/csrc/api.hip -o build/libcp_ws_$n.so || exit 1
  done
else
  for v in "${V[@]}"; do
    n=${v%%:*}
    echo "== build: $n (-DCP_VARIANTS -DWS_STAMP ${v#*:}) =="
    CPNATIVE_LIB=build/libcp_ws_$n.so WS_STAMP=1 timeout -k 10 120 python tools/ws_bench.py 2>&1 | grep -A40 "in-kernel stamps" || exit 1
  done
fi
