#!/bin/bash
# usage (GPU box): bash tools/probes/run_dec_gemm_pmc.sh -- PMC passes over the stand-alone decoder GEMM probe (LDS conflicts, MFMA busy, waits)
set -e -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_probe9; mkdir -p $O
cd $R/tools/probes
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-value -o /tmp/dec_gemm_bench dec_gemm_bench.hip
cd /tmp
for set in "lds:SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "wave:SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "vm:SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  n=${set%%:*}; c=${set#*:}
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- /tmp/dec_gemm_bench 4096 10000 > $O/pmc_$n.log 2>&1 || true
done
cd $R
for k in dec_gemm320_dma_kernel dec_gemm320_dma4_kernel; do
  { echo "# $k, stand-alone probe B=4096 G=10000 (both A orientations averaged)"; for d in lds mfma wave vm; do f=$(find $O/pmc_$d -name "*counter_collection.csv" | head -1); echo "== --pmc pass: $d"; [ -n "$f" ] && python tools/pmc_summary.py $f $k; done; } > $O/pmc_$k.txt
  cat $O/pmc_$k.txt
done
