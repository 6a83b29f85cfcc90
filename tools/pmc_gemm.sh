#!/bin/bash
# usage (GPU box): bash tools/pmc_gemm.sh <tag>  -> per-kernel PMC averages for the step's kernels (3 counter passes, eager bench)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${T}_pmc$i -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > $O/${T}_pmc$i.log 2>&1
done
cd $R
python tools/pmc_table.py $O/${T}_pmc1 $O/${T}_pmc2 $O/${T}_pmc3 > $O/${T}_pmc_table.txt
cat $O/${T}_pmc_table.txt
