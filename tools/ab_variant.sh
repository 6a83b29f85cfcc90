#!/bin/bash
# A/B of a compile-time kernel variant selected by an environment variable (dev tool): ab_variant.sh VAR "0 1 2" entry_point
for rep in 1 2; do
for v in $2; do
  echo -n "$1=$v  "
  env $1=$v python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pe=d['roofline']['per_entry_point']
print('step %.4f ms   %s %.1f us' % (d['ms_per_step'], '$3', pe['$3']['avg_ms']*1e3))"
done
done
