#!/bin/bash
OUT=gpurun_out/r05g; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "guide_folded or one_call" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-configs --no-legs > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-legs --no-alt > /dev/null 2>&1; cp /tmp/prof_f/*kernel_stats.csv $GRAFT_REPO_ROOT/$OUT/ 2>/dev/null; head -4 $GRAFT_REPO_ROOT/$OUT/trace_kernel_stats.csv | cut -c1-150
