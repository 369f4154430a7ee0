#!/bin/bash
# A/B of a pv_conv_sp.hip variant library on the conv configs: bash scripts/abl_conv.sh "<variant> ..." (variants built by scripts/mkvariant_file.sh; "base" = the product)
for round in 1 2; do
for v in $1; do
  if [ $v = base ]; then L=""; else L="PV_LIB_PATH=pyroved_amd/variants/lib_$v.so"; fi
  for c in C5 C4; do
  echo -n "$v $c: "
  env $L timeout 300 python bench.py --config $c --fused 2 --steps 40 --warmup 10 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>/dev/null | python scripts/benchline.py
  done
done
done
