#!/bin/bash
# usage (GPU box): bash tools/probes/wgrad_tiles.sh <outdir> "<bench args>" tile...   -- per-kernel time of the fc1 weight-gradient pair launch for each gene tile
# (SPV_FC1_WGRAD_TILE), single stream, rocprofv3 kernel statistics
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; ARGS=$2; shift 2
mkdir -p $O; cd /tmp
export SPV_SERIAL_STREAMS=1
for t in "$@"; do
  export SPV_FC1_WGRAD_TILE=$t
  rm -rf $O/prof_$t
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$t -- python3 $R/bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline --no-elbo-delta > $O/prof_$t.log 2>&1 || echo "tile $t failed"
  f=$(find $O/prof_$t -name "*kernel_stats.csv" | head -1)
  echo "tile $t: $(grep -E 'fc1_wgrad' $f | cut -d, -f1-5 | tr '\n' ' ')"
  rm -rf $O/prof_$t
done
