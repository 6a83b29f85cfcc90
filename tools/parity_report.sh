#!/bin/bash
# usage (GPU box): bash tools/parity_report.sh [tag]  -> gpurun_out/<tag>_fullsize_parity_report.txt (measured errors of a whole step at the BASELINE shapes vs the oracle)
T=${1:-r03}
set -e -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SPV_PARITY_REPORT_ONLY=1 timeout -k 10 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -s 2>&1 | tee gpurun_out/${T}_fullsize_parity_report.txt | grep -E "fullsize parity|grad |passed|failed|Error" 
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize_parity.py 2>&1 | tail -25
