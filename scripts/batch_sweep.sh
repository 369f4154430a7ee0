#!/bin/bash
# Batch sweep of the headline model (C2's iVAE 28x28 ['r','t']) on both decoder paths: bench.py lines -> one row per batch.
#   bash scripts/batch_sweep.sh [tag]
TAG=${1:-sweep}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for f in 3 2; do
  for B in 64 128 256 512 1024 2048 8192 32768; do
    echo -n "fused=$f B=$B: "
    timeout 600 python bench.py --config C2 --fused $f --batch $B --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>>$OUT/err.log | python scripts/benchline.py
  done
done | tee $OUT/sweep.txt
