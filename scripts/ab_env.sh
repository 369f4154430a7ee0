#!/bin/bash
# A/B of one environment switch of the experiments build on one box: scripts/ab_env.sh NAME "v1 v2" [bench args]
name=$1; vals=$2; shift; shift
args=${@:---steps 200 --warmup 5 --no-alt --no-configs --no-legs --no-cpu-baseline}
export PV_LIB_PATH=$PWD/pyroved_amd/libpyroved_amd_exp.so
for i in 1 2 3; do
  for v in $vals; do
    env $name=$v python bench.py $args 2>&1 | tail -1 | cut -c1-140 | sed "s/^/$name=$v /"
  done
done
