#!/bin/bash
# usage (GPU box): bash tools/checkpoint.sh <tag>   -- one evidence set (copy what is to be judged from gpurun_out/ into profiles/): GPU tests, bench lines of every BASELINE config,
# rocprofv3 kernel statistics (default and single-stream), PMC passes for the likelihood kernel and the fc1 DMA GEMM, step timeline
set -e -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q --ignore=_prev > $O/${T}_gpu_tests.txt 2>&1 || { tail -30 $O/${T}_gpu_tests.txt; exit 1; }
tail -2 $O/${T}_gpu_tests.txt
timeout -k 10 600 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
cat $O/${T}_bench.json | cut -c1-400
for c in c1 c3 c4 c5; do timeout -k 10 600 python bench.py --config $c > $O/${T}_bench_$c.json 2> $O/${T}_bench_$c.err || echo "$c failed"; done
# the headline workload in "fp32" (split-bf16) mode: the step time at the precision that needs no tolerance argument (VERDICT r02 missing #2)
timeout -k 10 600 python bench.py --precision fp32 --no-cpu-baseline > $O/${T}_bench_c2_fp32.json 2> $O/${T}_bench_c2_fp32.err || echo "c2 fp32 failed"
cd /tmp
# (a) default command: two-stream overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-elbo-delta > $O/${T}_prof.log 2>&1
# (b) every launch on one stream: per-kernel durations (what bench.py's roofline block reports)
export SPV_SERIAL_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_serial -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-elbo-delta > $O/${T}_prof_serial.log 2>&1
for set in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "valu:SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" "wave:SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  n=${set%%:*}; c=${set#*:}
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${T}_pmc_$n -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-elbo-delta --no-graph > $O/${T}_pmc_$n.log 2>&1 || true
done
unset SPV_SERIAL_STREAMS
cd $R
{ echo "# workload: c2 bf16 u16 B4096 G10000"; for d in fetch write valu wave mfma; do f=$(find $O/${T}_pmc_$d -name "*counter_collection.csv" | head -1); echo "== --pmc pass: $d"; [ -n "$f" ] && python tools/pmc_summary.py $f dec_nb_kernel; done; } > $O/${T}_pmc_dec_nb_kernel.txt
{ echo "# workload: c2 bf16 u16 B4096 G10000"; for d in fetch write valu wave mfma; do f=$(find $O/${T}_pmc_$d -name "*counter_collection.csv" | head -1); echo "== --pmc pass: $d"; [ -n "$f" ] && python tools/pmc_summary.py $f fc1_fwd_dma_pair_kernel; done; } > $O/${T}_pmc_fc1_fwd_dma_pair_kernel.txt
{ echo "# workload: c2 bf16 u16 B4096 G10000"; for d in fetch write valu wave mfma; do f=$(find $O/${T}_pmc_$d -name "*counter_collection.csv" | head -1); echo "== --pmc pass: $d"; [ -n "$f" ] && python tools/pmc_summary.py $f fc1_wgrad_dma_pair_kernel; done; } > $O/${T}_pmc_fc1_wgrad_dma_pair_kernel.txt
for k in dec_gemm320_dma4_kernel dec_logits_dma_kernel dec_heads_bwd_kernel dec_lse_kernel reduce_slabs_kernel adam_images_kernel; do
  { echo "# workload: c2 bf16 u16 B4096 G10000"; for d in fetch write valu wave mfma; do f=$(find $O/${T}_pmc_$d -name "*counter_collection.csv" | head -1); echo "== --pmc pass: $d"; [ -n "$f" ] && python tools/pmc_summary.py $f $k; done; } > $O/${T}_pmc_$k.txt
done
cat $O/${T}_pmc_fc1_fwd_dma_pair_kernel.txt
f=$(find $O/${T}_prof_serial -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_serial_rocprofv3_kernel_stats.csv; python tools/prof_summary.py $f 38 40 > $O/${T}_serial_rocprofv3_kernel_stats_summary.txt; head -24 $O/${T}_serial_rocprofv3_kernel_stats_summary.txt
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_rocprofv3_kernel_stats.csv; python tools/prof_summary.py $f 38 40 > $O/${T}_rocprofv3_kernel_stats_summary.txt
bash tools/timeline.sh $T > /dev/null 2>&1 || true
cp $O/${T}_timeline.txt $O/${T}_step_timeline.txt 2>/dev/null || true
SPV_PARITY_REPORT_ONLY=1 timeout -k 10 600 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -s 2>&1 | grep -E "fullsize parity|grad |passed|failed" > $O/${T}_fullsize_parity_report.txt || true
mkdir -p $O/${T}_graph_edges && timeout -k 10 300 python tools/graph_dump.py $O/${T}_graph_edges/shipped_single --config single > $O/${T}_graph_edges/dump.log 2>&1 || true
