#!/bin/bash
OUT=gpurun_out/r05n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "guide_folded or one_call or bf16_mode or golden" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for v in "" variants/lib_red16.so variants/lib_red64.so; do
  L=pyroved_amd/libpyroved_amd.so; [ -n "$v" ] && L=pyroved_amd/$v
  echo "== $L"
  (cd /tmp && export TMPDIR=/tmp && PV_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-configs --no-legs --no-alt 2>/dev/null | python $GRAFT_REPO_ROOT/scripts/benchline.py; grep "latent_bwd_reduce\|wgrad_small" /tmp/prof_x/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120; rm -rf /tmp/prof_x)
done 2>&1 | tee $OUT/red.txt
