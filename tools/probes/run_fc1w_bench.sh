#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")"
O=../../gpurun_out/r02_probe6; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o /tmp/fc1w_bench fc1w_bench.hip 2>/dev/null
for cfg in "4096 10000" "4096 20000" "1000 3001" "130 333"; do
  timeout -k 10 120 /tmp/fc1w_bench $cfg | tee -a $O/fc1w_bench.txt
done
