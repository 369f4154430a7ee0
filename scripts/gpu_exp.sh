#!/bin/bash
# kernel experiments: bench with variant libraries (pyroved_amd/variants/*.so) and ablation masks
OUT=gpurun_out/${1:-exp}; mkdir -p $OUT
run() { # label, lib, ablate
  PV_LIB_PATH=$2 PV_FD_ABLATE=$3 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print('%-28s ablate=%-3s ms/step %.4f kernel_ms %.4f' % ('$1', '$3', d['ms_per_step'], d['roofline']['kernel_ms']))
except Exception as e: print('$1 failed', e)" | tee -a $OUT/exp.log
}
for A in 0 15; do run base "" $A; done
for V in pyroved_amd/variants/*.so; do for A in 0 15; do run $(basename $V .so) $PWD/$V $A; done; done
