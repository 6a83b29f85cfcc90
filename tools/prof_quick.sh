#!/bin/bash
# usage (on the GPU box): bash tools/prof_quick.sh <tag>  -> gpurun_out/<tag>_summary.txt (rocprofv3 kernel stats of a short bench run)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/${T}_prof.log 2>&1
cd $R
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py $f 38 > $O/${T}_summary.txt
grep -v "poisson\|gamma_cuda" $O/${T}_summary.txt | head -70
