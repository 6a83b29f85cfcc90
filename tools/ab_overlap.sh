#!/bin/bash
# A/B of SPV_OVERLAP_SMALL on one box (dev tool)
for v in 1 0 1 0 1 0; do
  echo -n "overlap=$v  "
  SPV_OVERLAP_SMALL=$v python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
