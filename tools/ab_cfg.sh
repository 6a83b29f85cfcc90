#!/bin/bash
# usage (GPU box): bash tools/ab_cfg.sh VAR v1 v2 -- c1 c3 ...   -> alternating bench.py --config runs with VAR=v (2 rounds per config)
set -e -o pipefail
cd $GRAFT_REPO_ROOT
VAR=$1; shift
vals=()
while [ "$1" != "--" ]; do vals+=("$1"); shift; done
shift
for c in "$@"; do for i in 1 2; do for v in "${vals[@]}"; do
  env $VAR=$v timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-elbo-delta 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c $VAR=$v', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4))"
done; done; done
