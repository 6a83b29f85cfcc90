#!/bin/bash
# usage (GPU box): bash tools/sweep_switches.sh <outfile> "<bench args>" [rounds]   -- one bench.py run per schedule switch setting (screening; confirm with tools/ab.sh)
O=$1; ARGS=$2; N=${3:-1}
bash tools/ab.sh -n $N -a "$ARGS" "-" "SPV_OVERLAP_SMALL=0" "SPV_STAGGER=0" "SPV_STAGGER_BWD=1" "SPV_DA_FIRST=0" "SPV_DA_FIRST=2" "SPV_FC1_PAIR_SPLITS=0" "SPV_DEC_PAIR_SPLITS=0" \
  "SPV_GSPLIT_WANT=256" "SPV_GSPLIT_WANT=1024" "SPV_DEFER_BC=0" "SPV_FWD_GROUP_STREAMS=0" "SPV_BWD_GROUP_STREAMS=0" "SPV_WM_LATE=1" "SPV_WM_LATE=0" "-" 2>&1 | tee $O
