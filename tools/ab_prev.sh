#!/bin/bash
# usage (GPU box): bash tools/ab_prev.sh [-n ROUNDS] [-a "bench args"]   -> alternating bench.py runs of THIS tree and of the copy of an
# earlier commit under _prev/ (made on the build machine: git archive <commit> | tar -x -C _prev; cd _prev && python -m spvipes_amd.build):
# a same-box A/B of two code versions
set -o pipefail
cd "$GRAFT_REPO_ROOT"
ROUNDS=2; ARGS=""
while getopts "n:a:" o; do case $o in n) ROUNDS=$OPTARG;; a) ARGS=$OPTARG;; esac; done
for i in $(seq 1 "$ROUNDS"); do
  for d in . _prev; do
    (cd $d && timeout -k 10 300 python bench.py --no-cpu-baseline --no-elbo-delta $ARGS 2>>"$GRAFT_REPO_ROOT/gpurun_out/ab_stderr.log" | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$d]', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4), flush=True)") || echo "[$d] FAILED"
  done
done
