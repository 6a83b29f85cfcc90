#!/bin/bash
# A/B of one environment switch on one box (dev tool): ab_env.sh VAR
for v in 1 0 1 0 1 0; do
  echo -n "$1=$v  "
  env $1=$v python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
