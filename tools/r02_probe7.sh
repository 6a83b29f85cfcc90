#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_probe7; mkdir -p $O
cd $R/tools/probes
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o /tmp/nb_bench nb_bench.hip 2>/dev/null
for n in 1 2 1 2 4; do NB_CELL_TILES=$n timeout -k 10 120 /tmp/nb_bench 4096 10000 | sed "s/^/tiles=$n /" | tee -a $O/nb_tiles.txt; done
cd $R
bash tools/timeline.sh r02e > /dev/null 2>&1 || true
head -14 gpurun_out/r02e_timeline.txt
