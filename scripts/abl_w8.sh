#!/bin/bash
# A/B of -DW8_ABL timing-ablation builds of the 8-wave decoder kernel (scripts/mkvariant_file.sh w8abl<N> pv_sdec_fused_w8.hip -DW8_ABL=<N>)
#   bash scripts/abl_w8.sh "<variant> ..." "<batch> ..."
mkdir -p gpurun_out/abl
for round in 1 2; do
for v in $1; do
  if [ $v = base ]; then L=""; else L="PV_LIB_PATH=pyroved_amd/variants/lib_$v.so"; fi
  for B in $2; do
  echo -n "$v B=$B: "
  env $L timeout 300 python bench.py --config C2 --fused 3 --batch $B --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>>gpurun_out/abl/err.log | python scripts/benchline.py
  done
done
done | tee -a gpurun_out/abl/ab.txt
