#!/bin/bash
# fc1 forward probe: the 128-cell LDS-DMA kernel (the 256-cell variant was measured and removed: docs/lab_notes.md D), one group and a
# two-group grid, C2 and C3 gene counts
#   run_fc1_bench.sh <out dir under gpurun_out> ["PAIR SPLITS" ...]
set -e -o pipefail
cd "$(dirname "$0")"
O=../../gpurun_out/${1:-fc1_probe}; mkdir -p $O
shift || true
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o /tmp/fc1_bench fc1_bench.hip
if [ $# -eq 0 ]; then set -- "0 8" "1 4" "1 8"; fi
for G in 10000 20000; do
for v in "$@"; do
  read b c <<< "$v"
  F1_PAIR=$b timeout -k 10 120 /tmp/fc1_bench 4096 $G $c | tee -a $O/fc1_bench.txt
done
done
