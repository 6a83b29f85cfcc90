#!/bin/bash
# usage (GPU box): bash tools/serial_stats.sh <tag>  -> gpurun_out/<tag>_serial_summary.txt (per-kernel durations, one stream)
set -e -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd /tmp
export SPV_SERIAL_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_serial -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-elbo-delta > $O/${T}_prof_serial.log 2>&1
cd $R
f=$(find $O/${T}_prof_serial -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py $f 38 60 > $O/${T}_serial_summary.txt
