#!/bin/bash
# A/B on one box: working tree vs the committed package copied + built under _prev/ (dev tool).
# bench.py imports the package that sits next to it, so the "prev" leg runs a copy of bench.py placed in _prev/.
# Prepare with:  rm -rf _prev && mkdir _prev && git archive HEAD spvipes_amd include oracle | tar -x -C _prev && (cd _prev && python -m spvipes_amd.build)
cp bench.py _prev/bench.py
mkdir -p _prev/profiles
for v in cur prev cur prev cur prev; do
  echo -n "$v  "
  if [ $v = prev ]; then f=_prev/bench.py; else f=bench.py; fi
  python $f --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['per_entry_point']['spv_dec_nb_fwd']['avg_ms'])"
done
