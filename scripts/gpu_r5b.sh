#!/bin/bash
OUT=gpurun_out/r05i; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "guide_folded or one_call or bf16_mode" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for v in trace trace_rot; do PV_LIB_PATH=pyroved_amd/variants/lib_$v.so python scripts/gpu_trace_w8.py > $OUT/$v.txt 2>&1; tail -3 $OUT/$v.txt | head -2; done
PV_TRACE_FOLD=0 PV_LIB_PATH=pyroved_amd/variants/lib_trace.so python scripts/gpu_trace_w8.py > $OUT/trace_nofold.txt 2>&1; tail -2 $OUT/trace_nofold.txt | head -1
timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-configs --no-legs > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-400
