#!/bin/bash
# usage (GPU box): bash tools/checkpoint.sh <tag> [a|b|ab]   -- one evidence set (copy what is to be judged from gpurun_out/ into profiles/)
#   part a: GPU tests, bench lines of every BASELINE config (+ the headline workload in split-word mode), rocprofv3 kernel statistics (default
#           and single-stream), step timeline, full-size parity report, captured graph's node / edge dump
#   part b: tools/pmc_workload.sh on c2, c3, c5 and c4 (five --pmc passes + serial statistics each; the tables bench.py's roofline block reads)
set -e -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; PART=${2:-ab}
mkdir -p $O
cd $R
if [[ $PART == *a* ]]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q --ignore=_prev > $O/${T}_gpu_tests.txt 2>&1 || { tail -30 $O/${T}_gpu_tests.txt; exit 1; }
  tail -2 $O/${T}_gpu_tests.txt
  timeout -k 10 600 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
  cut -c1-400 $O/${T}_bench.json
  for c in c1 c3 c4 c5; do timeout -k 10 600 python bench.py --config $c > $O/${T}_bench_$c.json 2> $O/${T}_bench_$c.err || echo "$c failed"; done
  # the headline workload in "fp32" (split-bf16) mode: the step time at the precision that needs no tolerance argument
  timeout -k 10 600 python bench.py --precision fp32 --no-cpu-baseline > $O/${T}_bench_c2_fp32.json 2> $O/${T}_bench_c2_fp32.err || echo "c2 fp32 failed"
  cd /tmp
  # default command: two-stream overlap
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-elbo-delta > $O/${T}_prof.log 2>&1
  # every launch on one stream: per-kernel durations (what bench.py's roofline block reports)
  SPV_SERIAL_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_serial -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-elbo-delta > $O/${T}_prof_serial.log 2>&1
  cd $R
  f=$(find $O/${T}_prof_serial -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_serial_rocprofv3_kernel_stats.csv; python tools/prof_summary.py $f 38 40 > $O/${T}_serial_rocprofv3_kernel_stats_summary.txt; head -24 $O/${T}_serial_rocprofv3_kernel_stats_summary.txt
  f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_rocprofv3_kernel_stats.csv; python tools/prof_summary.py $f 38 40 > $O/${T}_rocprofv3_kernel_stats_summary.txt
  rm -rf $O/${T}_prof $O/${T}_prof_serial
  bash tools/timeline.sh $T > /dev/null 2>&1 || true
  cp $O/${T}_timeline.txt $O/${T}_step_timeline.txt 2>/dev/null || true
  rm -rf $O/${T}_trace
  SPV_PARITY_REPORT_ONLY=1 timeout -k 10 600 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -s 2>&1 | grep -E "fullsize parity|grad |passed|failed" > $O/${T}_fullsize_parity_report.txt || true
  mkdir -p $O/${T}_graph_edges && timeout -k 10 300 python tools/graph_dump.py $O/${T}_graph_edges/shipped_single --config single > $O/${T}_graph_edges/dump.log 2>&1 || true
fi
if [[ $PART == *b* ]]; then
  bash tools/pmc_workload.sh $T c2
  bash tools/pmc_workload.sh $T c3
  bash tools/pmc_workload.sh $T c5
  bash tools/pmc_workload.sh $T c4
fi
