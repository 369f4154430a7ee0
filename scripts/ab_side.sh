#!/bin/bash
# A/B of the two-stream conv steps on the GPU box: bench lines of C5 / C4 with the side stream off, on, and on with paired launches.
#   bash scripts/ab_side.sh <tag>
TAG=${1:-abside}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() {  # name, env..., config, fused
  local name=$1; shift
  for cfg in C5 C4; do for f in 2 3; do
    echo -n "$name $cfg fused=$f: "
    env "$@" timeout 300 python bench.py --config $cfg --fused $f --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>>$OUT/err.log | python scripts/benchline.py
  done; done
}
run off PV_NO_SIDE=1 | tee -a $OUT/ab.txt
run on PV_X=0 | tee -a $OUT/ab.txt
run on+pairs PV_SIDE_PAIR=1 | tee -a $OUT/ab.txt
run off PV_NO_SIDE=1 | tee -a $OUT/ab.txt
run on PV_X=0 | tee -a $OUT/ab.txt
