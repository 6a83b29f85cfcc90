set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/r01i_gpu_tests.txt 2>&1 || { tail -20 $O/r01i_gpu_tests.txt; exit 1; }
tail -2 $O/r01i_gpu_tests.txt
python bench.py > $O/r01i_bench.json 2> $O/r01i_bench.err
cat $O/r01i_bench.json
cd /tmp
# (a) the default command: step timing with the two-stream overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r01i_prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/r01i_prof.log 2>&1
# (b) the same benchmark with every launch on one stream: per-kernel durations (what bench.py's roofline block reports)
export SPV_SERIAL_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r01i_prof_serial -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/r01i_prof_serial.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r01i_pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $O/r01i_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r01i_pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $O/r01i_pmc_write.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/r01i_pmc_valu -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $O/r01i_pmc_valu.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/r01i_pmc_wave -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > $O/r01i_pmc_wave.log 2>&1 || true
unset SPV_SERIAL_STREAMS
cd $R
for d in fetch write valu wave; do f=$(find $O/r01i_pmc_$d -name "*counter_collection.csv" | head -1); echo "== $d $f"; [ -n "$f" ] && python tools/pmc_summary.py $f dec_nb_kernel; done
f=$(find $O/r01i_prof_serial -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py $f 38 40 > $O/r01i_serial_summary.txt; head -16 $O/r01i_serial_summary.txt
f=$(find $O/r01i_prof -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py $f 38 40 > $O/r01i_summary.txt
bash tools/timeline.sh r01i > /dev/null 2>&1 || true
