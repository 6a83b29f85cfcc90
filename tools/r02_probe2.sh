#!/bin/bash
# round-2 probe 2: ablations of the likelihood kernel (what are its waves waiting for?)
set -e -o pipefail
cd "$(dirname "$0")/probes"
O=../../gpurun_out/r02_probe2; mkdir -p $O
for a in 0 1 2 4 8 16 3 7 31; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DSPV_NB_ABLATE=$a -o /tmp/nb_a$a nb_bench.hip 2>/dev/null
  echo "ABLATE=$a" | tee -a $O/nb_ablate.txt
  timeout -k 10 120 /tmp/nb_a$a 4096 10000 68.1e6 | tee -a $O/nb_ablate.txt
done
