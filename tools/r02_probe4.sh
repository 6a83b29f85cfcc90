#!/bin/bash
# round-2 probe 4: LDS-DMA fc1 forward kernel (correctness vs naive + timing), then the GPU test-suite
set -e -o pipefail
cd "$(dirname "$0")/probes"
O=../../gpurun_out/r02_probe4; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o /tmp/fc1_bench fc1_bench.hip 2>/dev/null
for cfg in "4096 10000 8" "4096 10000 4" "4096 10000 16" "4096 20000 8" "1000 3001 0"; do
  timeout -k 10 120 /tmp/fc1_bench $cfg | tee -a $O/fc1_bench.txt
done
cd ../..
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
