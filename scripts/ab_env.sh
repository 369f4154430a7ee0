#!/bin/bash
# A/B of environment knobs on the GPU box (round 5: the knobs exist in the experiments build only — loaded here through PV_LIB_PATH): bench lines of the given configs under each "NAME=VALUE[,NAME=VALUE...]" setting ("-" = defaults).
#   bash scripts/ab_env.sh <tag> "<cfg> ..." "<fused> ..." <setting> [<setting> ...]
TAG=$1; CFGS=$2; FUSED=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
  for setting in "$@"; do
    envs=(); [ "$setting" != "-" ] && IFS=',' read -ra envs <<< "$setting"
    for cfg in $CFGS; do for f in $FUSED; do
      echo -n "$setting $cfg fused=$f: "
      env PV_LIB_PATH=${PV_LIB_PATH:-pyroved_amd/libpyroved_amd_exp.so} "${envs[@]}" timeout 300 python bench.py --config $cfg --fused $f --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>>$OUT/err.log | python scripts/benchline.py
    done; done
  done
done | tee $OUT/ab.txt
