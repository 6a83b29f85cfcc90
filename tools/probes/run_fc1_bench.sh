#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")"
O=../../gpurun_out/r02_probe5; mkdir -p $O
for pp in 0 1; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DF1_PIPE=$pp -o /tmp/fc1_bench_$pp fc1_bench.hip 2>/dev/null
echo "F1_PIPE=$pp" | tee -a $O/fc1_bench.txt
for cfg in "4096 10000 8" "4096 20000 8"; do
  timeout -k 10 120 /tmp/fc1_bench_$pp $cfg | tee -a $O/fc1_bench.txt
  F1_DISTINCT_ROWS=128 timeout -k 10 120 /tmp/fc1_bench_$pp $cfg | tee -a $O/fc1_bench.txt
done
done
