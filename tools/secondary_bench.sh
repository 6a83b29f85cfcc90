#!/bin/bash
# secondary workloads quoted in DESIGN.md section 5 (dev tool): C1, the C3 per-GPU shard shape, fp32 mode, fp32-stored counts
run() { echo -n "$1: "; shift; python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step', round(d['value']/1e6,3), 'M cells/s  loss', d['final_loss'])"; }
run "C1 (2x2000x2000, B=128, H=64)" --cells 2000 --genes 2000 --batch-size 128 --n-hidden 64 --steps 100
run "C3 shard (G=20000)" --genes 20000 --cells 30000
run "fp32 mode" --precision fp32
run "fp32-stored counts" --count-dtype f32 --cells 30000
run "C2 default" 
