#!/bin/bash
OUT=gpurun_out/r05r; mkdir -p $OUT
export PV_LIB_PATH=$PWD/pyroved_amd/variants/lib_dg1.so
for d in seed0 seed7 blobs; do PV_DRAW=$d PV_THREADS=32 timeout 900 python scripts/grad_margin.py C4 C5 > $OUT/grad_margin_$d.txt 2>&1; grep "smallest\|^==" $OUT/grad_margin_$d.txt; done
for c in C5 C4; do echo -n "$c dg1: "; timeout 300 python bench.py --config $c --fused 2 --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>/dev/null | python scripts/benchline.py; done
unset PV_LIB_PATH
for c in C5 C4; do echo -n "$c dg2: "; timeout 300 python bench.py --config $c --fused 2 --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>/dev/null | python scripts/benchline.py; done
