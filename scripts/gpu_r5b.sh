#!/bin/bash
OUT=gpurun_out/r05u; mkdir -p $OUT
for d in seed0 seed7 blobs; do PV_DRAW=$d PV_THREADS=32 timeout 900 python scripts/grad_margin.py C4 > $OUT/grad_margin_$d.txt 2>&1; grep "smallest\|^==" $OUT/grad_margin_$d.txt; grep "decoder\|fc_latent" $OUT/grad_margin_$d.txt | cut -c1-100; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "full_size_c4 or convenc or 64x64" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for c in C4 C4fc; do echo -n "$c: "; timeout 300 python bench.py --config $c --fused 2 --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>/dev/null | python scripts/benchline.py; done
