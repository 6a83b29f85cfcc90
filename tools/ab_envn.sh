#!/bin/bash
# usage (GPU box): bash tools/ab_envn.sh VAR v1 v2 [v3 ...]   -> alternating bench.py runs with VAR=v (3 rounds)
set -e -o pipefail
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for i in 1 2 3; do for v in "$@"; do env $VAR=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-elbo-delta 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4))"; done; done
