#!/bin/bash
# A/B of two builds of the library on bench configs: bash scripts/ab_lib.sh <other.so> C5 [C4 ...]   (PV_LIB_PATH selects the build)
OTHER=$1; shift
for CFG in "$@"; do
  for lib in "" "$OTHER" "" "$OTHER"; do
    env PV_LIB_PATH=$lib python bench.py --config $CFG --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); f=d.get('fp32_class') or {}
print('$CFG', '${lib:-current}', d['dtype'], round(d['ms_per_step'],4), f.get('dtype'), round(f.get('ms_per_step',0),4))"
  done
done
