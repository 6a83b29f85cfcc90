#!/bin/bash
# round-2 probe 1: VALU issue microbench + likelihood kernel occupancy sweep (built on the GPU box)
set -e -o pipefail
cd "$(dirname "$0")/probes"
O=../../gpurun_out/r02_probe1; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue valu_issue.hip 2>/dev/null
timeout -k 10 120 /tmp/valu_issue | tee $O/valu_issue.txt
for o in 2 3 4; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DSPV_NB_OCC=$o -o /tmp/nb_bench_$o nb_bench.hip 2>/dev/null
  timeout -k 10 120 /tmp/nb_bench_$o 4096 10000 68.1e6 | tee -a $O/nb_bench.txt
done
