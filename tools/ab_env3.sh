#!/bin/bash
# usage (GPU box): bash tools/ab_env3.sh VAR   -> alternating A/B of bench.py with VAR=1 / VAR=0 (3 rounds)
set -e -o pipefail
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in 1 0; do env $1=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-elbo-delta 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1=$v', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4))"; done; done
