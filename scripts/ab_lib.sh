#!/bin/bash
# A/B of two builds of the library on one box: scripts/ab_lib.sh <libA> <libB> [bench args]   (fp32-class C2 step by default)
a=$1; b=$2; shift; shift
args=${@:---steps 100 --warmup 5 --fused 2 --no-alt --no-configs --no-legs --no-cpu-baseline}
for i in 1 2 3; do
  for l in $a $b; do
    PV_LIB_PATH=$PWD/$l python bench.py $args 2>&1 | tail -1 | sed "s|^|$(basename $l) |" | cut -c1-200
  done
done
