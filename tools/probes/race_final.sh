#!/bin/bash
# dev probe driver: the shipped build, whole training step, both precisions, under GPU sharing
export RACE_NOISE_S=150
echo "shipped build, fp32 mode, whole step"; timeout -k 10 400 python tools/probes/race_hunt.py 0 fp32 2500 2>&1 | grep -v amdgpu | tail -1
export RACE_NOISE_S=100
echo "shipped build, bf16 mode, whole step"; timeout -k 10 300 python tools/probes/race_hunt.py 0 bf16 1500 2>&1 | grep -v amdgpu | tail -1
echo "shipped build, fp32 mode, split backward"; timeout -k 10 300 python tools/probes/race_hunt.py 1 fp32 1500 2>&1 | grep -v amdgpu | tail -1
if [ -d _prev2 ]; then
  echo "forcezero + no SLP, bf16 mode, forward"; RACE_FWD_ONLY=1 timeout -k 10 300 python _prev2/tools/probes/race_hunt.py 0 bf16 1500 2>&1 | grep -v amdgpu | tail -1
  echo "forcezero + no SLP, fp32 mode, whole step"; timeout -k 10 300 python _prev2/tools/probes/race_hunt.py 0 fp32 1500 2>&1 | grep -v amdgpu | tail -1
fi
