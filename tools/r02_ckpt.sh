#!/bin/bash
# usage (GPU box): bash tools/r02_ckpt.sh <tag> [quick]   -> GPU tests + bench (+ rocprof summaries unless quick) into gpurun_out/<tag>_*
set -e -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.txt 2>&1 || { tail -40 $O/${T}_gpu_tests.txt; exit 1; }
tail -3 $O/${T}_gpu_tests.txt
timeout -k 10 600 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err || { tail -20 $O/${T}_bench.err; exit 1; }
cat $O/${T}_bench.json
