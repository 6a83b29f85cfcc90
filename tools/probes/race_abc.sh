#!/bin/bash
# dev probe driver: forcezero builds under _prev (SLP on) and _prev2 (SLP off); see DESIGN.md section 8
export RACE_FWD_ONLY=1
export RACE_NOISE_S=100
echo "A: SLP on, bf16 mode";  timeout -k 10 300 python _prev/tools/probes/race_hunt.py 0 bf16 1500 2>&1 | grep -v amdgpu | tail -1
echo "B: SLP on, fp32 mode";  timeout -k 10 300 python _prev/tools/probes/race_hunt.py 0 fp32 1000 2>&1 | grep -v amdgpu | tail -1
export RACE_NOISE_S=160
echo "C: SLP off, fp32 mode"; timeout -k 10 400 python _prev2/tools/probes/race_hunt.py 0 fp32 3000 2>&1 | grep -v amdgpu | tail -1
