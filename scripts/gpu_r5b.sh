#!/bin/bash
# scratch: kernel averages of a config under settings of the experiments build
OUT=gpurun_out/r05ak; mkdir -p $OUT
for setting in "PV_NO_SIDE=1" "PV_NO_SIDE=1,PV_K1_NOWIDE=1" "PV_NO_SIDE=1,PV_K1_WIDEB=2"; do
  envs=(); [ "$setting" != "-" ] && IFS=',' read -ra envs <<< "$setting"
  echo "== $setting"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_x && env PV_LIB_PATH=$GRAFT_REPO_ROOT/pyroved_amd/libpyroved_amd_exp.so "${envs[@]}" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o trace -- python $GRAFT_REPO_ROOT/bench.py --config C5 --fused 2 --steps 30 --warmup 10 --repeats 2 --no-cpu-baseline --no-configs --no-legs --no-alt 2>/dev/null | python $GRAFT_REPO_ROOT/scripts/benchline.py; python $GRAFT_REPO_ROOT/scripts/kstats.py /tmp/prof_x/*kernel_stats.csv k1_wgrad finish_table dec1d)
done 2>&1 | tee $OUT/out.txt
