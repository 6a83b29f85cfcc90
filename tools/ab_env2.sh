#!/bin/bash
# A/B of one environment switch with a per-entry-point time (dev tool): ab_env2.sh VAR "extra env" entry_point
for v in 1 0 1 0; do
  echo -n "$1=$v $2 "
  env $1=$v $2 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pe=d['roofline']['per_entry_point']
print('step %.4f ms  ' % d['ms_per_step'], '  '.join('%s %.1f' % (k[4:], v['avg_ms']*1e3) for k,v in pe.items()))"
done
