#!/bin/bash
# usage (GPU box): bash tools/ep_times.sh VAR v1 v2 ...  -> per-entry-point event times of bench.py's eager pass for each value of VAR
set -e -o pipefail
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do env $VAR=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-elbo-delta --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],4), {k: round(v['avg_ms']*1e3,1) for k,v in d['roofline']['per_entry_point'].items()})"; done
