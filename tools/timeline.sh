#!/bin/bash
# usage (on the GPU box): bash tools/timeline.sh <tag>  -> gpurun_out/<tag>_timeline.txt (kernel start/end of one replayed step)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/${T}_trace -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/${T}_trace.log 2>&1
cd $R
f=$(find $O/${T}_trace -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f > $O/${T}_timeline.txt 2>&1 || true
python tools/trace_timeline.py $f >> $O/${T}_timeline.txt
