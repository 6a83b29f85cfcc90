#!/bin/bash
# A/B on one box: working tree vs the committed package under _prev/ (dev tool)
for v in cur prev cur prev cur prev; do
  echo -n "$v  "
  if [ $v = prev ]; then P=_prev; else P=.; fi
  PYTHONPATH=$P python -c "
import sys, runpy, io, json, contextlib
sys.path.insert(0, '$P')
sys.argv = ['bench.py', '--no-cpu-baseline', '--steps', '60']
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path('bench.py', run_name='__main__')
print(json.loads(buf.getvalue().strip().splitlines()[-1])['ms_per_step'])
" 2>/dev/null
done
