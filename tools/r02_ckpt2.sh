#!/bin/bash
# GPU tests + default bench + A/B of SPV_DZ_ONLY + the other BASELINE configs
set -e -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.txt 2>&1 || { tail -40 $O/${T}_gpu_tests.txt; exit 1; }
tail -3 $O/${T}_gpu_tests.txt
for v in 1 0 1 0; do SPV_DZ_ONLY=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-elbo-delta 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DZ_ONLY=$v', d['ms_per_step'], d['ms_per_step_median'])"; done
timeout -k 10 600 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err || { tail -20 $O/${T}_bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/${T}_bench.json')); print({k: d[k] for k in ('value','ms_per_step','ms_per_step_median','elbo_delta')})"
for c in c3 c4 c5 c1; do timeout -k 10 600 python bench.py --config $c --no-cpu-baseline > $O/${T}_bench_$c.json 2> $O/${T}_bench_$c.err || { echo "$c failed"; tail -5 $O/${T}_bench_$c.err; }; python -c "import json; d=json.load(open('$O/${T}_bench_$c.json')); print('$c', {k: d[k] for k in ('value','ms_per_step','ms_per_step_median','elbo_delta')})" || true; done
