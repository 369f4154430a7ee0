#!/bin/bash
# A/B of environment settings on a bench config: bash scripts/ab_env.sh C5 "PV_K1_NB=0" "PV_K1_NB=2" ...  (each setting run twice, interleaved)
CFG=$1; shift
for rep in 1 2; do
  for kv in "$@"; do
    env $kv python bench.py --config $CFG --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); f=d.get('fp32_class') or {}
print('$CFG', '$kv', d['dtype'], round(d['ms_per_step'],4), f.get('dtype'), round(f.get('ms_per_step',0),4))"
  done
done
