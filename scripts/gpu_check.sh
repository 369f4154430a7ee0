#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, a short bench and a rocprofv3 kernel trace.
# Usage: scripts/gpu_check.sh [tag] [pytest-args...]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).total_memory>>30,'GiB')" > $OUT/env.log 2>&1
nproc >> $OUT/env.log; lscpu | grep "Model name" >> $OUT/env.log
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
echo "== pytest"; timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
echo "== bench"; timeout 900 python bench.py --steps 50 --warmup 10 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/bench.log
tail -2 $OUT/bench.log
echo "== rocprof"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1; echo "rocprof rc=$?")
find /tmp/prof_$TAG -name "*stats*.csv" -exec cp {} $OUT/ \; 2>/dev/null
find /tmp/prof_$TAG -type f | head -20
ls -la $OUT
echo "== bench fused=2"; timeout 600 python bench.py --steps 50 --warmup 10 --fused 2 --no-cpu-baseline > $OUT/bench_f2.log 2>&1; tail -1 $OUT/bench_f2.log | cut -c1-400
