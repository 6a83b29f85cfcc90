#!/bin/bash
# dev probe driver: SLP-on forcezero builds of the sources at commit 6212ed5 (before the arithmetic of the likelihood kernel was trimmed; `git archive 6212ed5 spvipes_amd include`), without (_prev) and with
# (_prev2) `-mllvm -amdgpu-snop-padding=2` -- does spacing every instruction out remove the packed-fp32 symptom?
export RACE_FWD_ONLY=1
export RACE_NOISE_S=100
echo "SLP on, forcezero, bf16";                 timeout -k 10 300 python _prev/tools/probes/race_hunt.py 0 bf16 1500 2>&1 | grep -v amdgpu | tail -1
echo "SLP on, forcezero + s_nop padding, bf16"; timeout -k 10 400 python _prev2/tools/probes/race_hunt.py 0 bf16 1500 2>&1 | grep -v amdgpu | tail -1
