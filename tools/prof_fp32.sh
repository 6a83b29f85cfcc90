set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp32_prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision fp32 > $O/fp32_prof.log 2>&1
cd $R
f=$(find $O/fp32_prof -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py $f 38 > $O/fp32_summary.txt
grep -v "poisson\|gamma_cuda" $O/fp32_summary.txt | head -40 | cut -c1-150
