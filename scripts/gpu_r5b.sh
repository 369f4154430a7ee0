#!/bin/bash
OUT=gpurun_out/r05k; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "guide_folded or one_call or bf16_mode" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
PV_LIB_PATH=pyroved_amd/variants/lib_trace.so python scripts/gpu_trace_w8.py > $OUT/trace.txt 2>&1; tail -3 $OUT/trace.txt | head -2
timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-configs --no-legs > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-400
