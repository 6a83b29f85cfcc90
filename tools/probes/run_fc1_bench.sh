#!/bin/bash
# fc1 forward probe: the 128-cell and the 256-cell LDS-DMA kernels, one group and a two-group grid, C2 and C3 gene counts
#   run_fc1_bench.sh <out dir under gpurun_out> ["BM256 PAIR SPLITS" ...]
set -e -o pipefail
cd "$(dirname "$0")"
O=../../gpurun_out/${1:-fc1_probe}; mkdir -p $O
shift || true
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o /tmp/fc1_bench fc1_bench.hip
if [ $# -eq 0 ]; then set -- "0 0 8" "1 0 16" "0 1 8" "1 1 8" "1 1 16"; fi
for G in 10000 20000; do
for v in "$@"; do
  read a b c <<< "$v"
  F1_BM256=$a F1_PAIR=$b timeout -k 10 120 /tmp/fc1_bench 4096 $G $c | tee -a $O/fc1_bench.txt
done
done
