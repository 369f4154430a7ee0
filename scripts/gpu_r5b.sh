#!/bin/bash
# scratch: isolated (one-stream) kernel averages of C5 and C4
OUT=gpurun_out/r05ap; mkdir -p $OUT
for cfg in C5 C4; do
  echo "== $cfg one stream"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_x && env PV_LIB_PATH=$GRAFT_REPO_ROOT/pyroved_amd/libpyroved_amd_exp.so PV_NO_SIDE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o trace -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --fused 2 --steps 30 --warmup 10 --repeats 2 --no-cpu-baseline --no-configs --no-legs --no-alt 2>/dev/null | python $GRAFT_REPO_ROOT/scripts/benchline.py; python $GRAFT_REPO_ROOT/scripts/kstats.py /tmp/prof_x/*kernel_stats.csv | head -24)
done 2>&1 | tee $OUT/out.txt
