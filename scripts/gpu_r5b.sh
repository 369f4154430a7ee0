#!/bin/bash
# round 5: gradient margins on the three draws + parity of the column-parallel tail + a bench A/B
OUT=gpurun_out/r05b; mkdir -p $OUT
for d in seed0 seed7 blobs; do PV_DRAW=$d PV_THREADS=32 timeout 900 python scripts/grad_margin.py C4 C5 > $OUT/grad_margin_$d.txt 2>&1; done
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "w8 or bf16_mode or one_call or full_size_properties or golden or weight_range or fused_forward" > $OUT/pytest_w8.log 2>&1; tail -5 $OUT/pytest_w8.log
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-configs --no-legs > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-600
