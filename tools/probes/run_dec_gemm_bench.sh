#!/bin/bash
# usage (GPU box): bash tools/probes/run_dec_gemm_bench.sh  -- LDS-DMA 320-column decoder GEMMs alone (tools/probes/dec_gemm_bench.hip)
set -e -o pipefail
cd $GRAFT_REPO_ROOT/tools/probes
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r02_probe8
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-value -o /tmp/dec_gemm_bench dec_gemm_bench.hip
{ timeout -k 10 120 /tmp/dec_gemm_bench 130 333 && timeout -k 10 120 /tmp/dec_gemm_bench 1000 3001 && timeout -k 10 120 /tmp/dec_gemm_bench 4096 10000 && timeout -k 10 120 /tmp/dec_gemm_bench 4096 20000; } 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r02_probe8/dec_gemm_bench.txt
