#!/bin/bash
# usage (GPU box): bash tools/ab.sh [-n ROUNDS] [-a "bench args"] "A=1 B=2" "A=0" ...   -> alternating bench.py runs, one per environment
# setting string (use "-" for the default environment), ROUNDS times round-robin: same box, so the numbers are comparable
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
ROUNDS=2; ARGS=""
while getopts "n:a:" o; do case $o in n) ROUNDS=$OPTARG;; a) ARGS=$OPTARG;; esac; done
shift $((OPTIND - 1))
for i in $(seq 1 "$ROUNDS"); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-elbo-delta $ARGS 2>>"$GRAFT_REPO_ROOT/gpurun_out/ab_stderr.log" | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4), flush=True)" || echo "[$v] FAILED"
  done
done
