#!/bin/bash
# round-2 probe 3: phase stamps of the likelihood kernel + dependent-chain issue rates
set -e -o pipefail
cd "$(dirname "$0")/probes"
O=../../gpurun_out/r02_probe3; mkdir -p $O
for a in 0 31; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DSPV_NB_STAMPS -DSPV_NB_ABLATE=$a -o /tmp/nb_st nb_bench.hip 2>/dev/null
timeout -k 10 120 /tmp/nb_st 4096 10000 68.1e6 | tee -a $O/nb_stamps.txt
done
