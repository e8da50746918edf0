#!/bin/bash
# build/libcp_prev.so = the library as of git revision $1 (default HEAD), for tools/ab_bench.sh build/libcp_prev.so (A/B on one box)
REV=${1:-HEAD}
cd "$(dirname "$0")/.."
rm -rf /tmp/prev_src && mkdir -p /tmp/prev_src/contrastiveprosthetics_amd /tmp/prev_src/include build
git archive $REV contrastiveprosthetics_amd/csrc include | tar -x -C /tmp/prev_src
(cd /tmp/prev_src/contrastiveprosthetics_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared api.hip -o /tmp/prev_src/libcp_prev.so) && cp /tmp/prev_src/libcp_prev.so build/libcp_prev.so
