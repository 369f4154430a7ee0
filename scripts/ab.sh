#!/bin/bash
# A/B of one environment knob on a bench config: bash scripts/ab.sh C5 PV_NO_K1   (prints ms/step of both legs, knob off / on)
CFG=${1:-C5}; KNOB=${2:-PV_NO_K1}
for v in 0 1 0 1; do
  env $KNOB=$v python bench.py --config $CFG --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); f=d.get('fp32_class') or {}
print('$KNOB=$v', d['dtype'], round(d['ms_per_step'],4), f.get('dtype'), round(f.get('ms_per_step',0),4))"
done
