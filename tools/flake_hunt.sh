#!/bin/bash
# usage (GPU box): bash tools/flake_hunt.sh N [ENV=VAL ...]  -- run the two-rank graph test N times, count failures
N=$1; shift
fails=0
for i in $(seq 1 $N); do
  if ! env "$@" timeout -k 10 200 python -m pytest "tests/test_gpu_dp_overlap.py::test_two_ranks_reduce_to_the_mean_gradient_and_stay_identical" -x -q > /tmp/fh_$i.txt 2>&1; then
    fails=$((fails+1)); cp /tmp/fh_$i.txt $GRAFT_REPO_ROOT/gpurun_out/flake_fail_$i.txt
  fi
done
echo "failures: $fails of $N ($*)"
