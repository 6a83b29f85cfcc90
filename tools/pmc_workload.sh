#!/bin/bash
# usage (GPU box): bash tools/pmc_workload.sh <tag> <preset c1..c5> ["extra bench.py args"]
# One counter evidence set for ONE bench workload: five separate rocprofv3 --pmc passes (never combined with trace domains other than
# --kernel-trace) of `bench.py --config <preset> --steps 4 --warmup 1 --no-graph` on one stream, plus the single-stream kernel statistics of
# the same command.  Writes gpurun_out/<tag>_<preset>_pmc_table.{json,txt} (every kernel, per launch; the `workload` tag inside is what
# bench.py matches) and gpurun_out/<tag>_<preset>_serial_rocprofv3_kernel_stats_summary.txt.  Copy what is to be judged into profiles/.
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; C=$2; X=$3
mkdir -p $O
WL=$(cd $R && python - <<PY
import bench
a = bench.parse("--config $C $X".split())
print(f"{a.config} {a.precision} {a.count_dtype} B{a.batch_size} G{a.genes}")
PY
)
echo "workload: $WL"
cd /tmp
export SPV_SERIAL_STREAMS=1
STEPS=4; WARM=1
# eager steps the profiler sees per pass: elbo-delta off, warm-up + timed (+ none of the event pass: --no-graph)
for set in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "valu:SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" "wave:SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  n=${set%%:*}; c=${set#*:}
  rm -rf $O/${T}_${C}_pmc_$n
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${T}_${C}_pmc_$n -- python3 $R/bench.py --config $C $X --steps $STEPS --warmup $WARM --no-cpu-baseline --no-elbo-delta --no-graph > $O/${T}_${C}_pmc_$n.log 2>&1 || echo "pass $n failed"
done
rm -rf $O/${T}_${C}_prof_serial
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_${C}_prof_serial -- python3 $R/bench.py --config $C $X --steps 10 --warmup 3 --no-cpu-baseline --no-elbo-delta > $O/${T}_${C}_prof_serial.log 2>&1 || echo "serial stats failed"
unset SPV_SERIAL_STREAMS
cd $R
python tools/pmc_json.py "$WL" $((STEPS + WARM)) $O/${T}_${C}_pmc_table.json $O/${T}_${C}_pmc_table.txt $O/${T}_${C}_pmc_fetch $O/${T}_${C}_pmc_write $O/${T}_${C}_pmc_valu $O/${T}_${C}_pmc_wave $O/${T}_${C}_pmc_mfma
f=$(find $O/${T}_${C}_prof_serial -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python tools/prof_summary.py $f 28 34 > $O/${T}_${C}_serial_rocprofv3_kernel_stats_summary.txt
head -14 $O/${T}_${C}_pmc_table.txt | cut -c1-200
# the raw pass directories are large: keep the tables only
rm -rf $O/${T}_${C}_pmc_fetch $O/${T}_${C}_pmc_write $O/${T}_${C}_pmc_valu $O/${T}_${C}_pmc_wave $O/${T}_${C}_pmc_mfma $O/${T}_${C}_prof_serial
