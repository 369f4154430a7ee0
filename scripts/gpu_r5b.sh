#!/bin/bash
# scratch: the record-reduction launch under variant builds (rocprof averages)
OUT=gpurun_out/r05ac; mkdir -p $OUT
for v in "" red16 red32 "" red16 red32; do
  lib=pyroved_amd/libpyroved_amd.so; [ -n "$v" ] && lib=pyroved_amd/variants/lib_$v.so
  (cd /tmp && export TMPDIR=/tmp && PV_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --repeats 2 --no-cpu-baseline --no-configs --no-legs --no-alt 2>/dev/null | python $GRAFT_REPO_ROOT/scripts/benchline.py; python $GRAFT_REPO_ROOT/scripts/kstats.py /tmp/prof_$v/*kernel_stats.csv latent_bwd_reduce wgrad_small sdec_w8)
done 2>&1 | tee $OUT/out.txt
