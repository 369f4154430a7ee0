#!/bin/bash
OUT=gpurun_out/r05x; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "first_block" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for c in C4 C5; do
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -o trace -- python $GRAFT_REPO_ROOT/bench.py --config $c --fused 2 --steps 30 --warmup 10 --repeats 2 --no-cpu-baseline --no-configs --no-legs --no-alt 2>/dev/null | python $GRAFT_REPO_ROOT/scripts/benchline.py; grep "c1_convpool" /tmp/prof_$c/*kernel_stats.csv | cut -c1-110)
done
