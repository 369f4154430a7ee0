#!/bin/bash
# Round-6 artefact run (as rounds 4 and 5) on the GPU box (from the repo root): bench line, rocprofv3 kernel stats of the same command, step
# timelines (C2 bf16 / fp32-class, C4, C5), SQ counters of the headline decoder kernels, and FETCH_SIZE / WRITE_SIZE passes of
# every config's dominant kernel(s) -> traffic.json (scripts/pmc_traffic.py).  Usage: bash scripts/gpu_prof_r6.sh <tag>
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
echo "== bench"; timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python scripts/benchline.py < $OUT/bench.json
echo "== rocprof kernel stats (the default bench command's headline legs)"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python $R/bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-configs --no-legs > $R/$OUT/rocprof.log 2>&1)
cp /tmp/prof_$TAG/trace_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
echo "== timelines"
BENCH_ARGS="--no-alt --no-configs --no-legs" bash scripts/gpu_timeline.sh ${TAG}_tl > /dev/null 2>&1; cp gpurun_out/${TAG}_tl/timeline.txt $OUT/step_timeline_bf16.txt
BENCH_ARGS="--no-alt --no-configs --no-legs --fused 2" bash scripts/gpu_timeline.sh ${TAG}_tl2 > /dev/null 2>&1; cp gpurun_out/${TAG}_tl2/timeline.txt $OUT/step_timeline_fp32class.txt
bash scripts/gpu_timeline_cfg.sh ${TAG}_c5tl C5 --no-configs --no-alt --fused 2 > /dev/null 2>&1; cp gpurun_out/${TAG}_c5tl/timeline.txt $OUT/ved_b256_step_timeline.txt
bash scripts/gpu_timeline_cfg.sh ${TAG}_c4tl C4 --no-configs --no-alt --fused 2 > /dev/null 2>&1; cp gpurun_out/${TAG}_c4tl/timeline.txt $OUT/ivae64_convenc_b128_step_timeline.txt
for c in C5 C4; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$c -o trace -- python $R/bench.py --config $c --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-configs > /dev/null 2>&1)
  cp /tmp/prof_${TAG}_$c/trace_kernel_stats.csv $OUT/${c}_kernel_stats.csv 2>/dev/null
done
echo "== SQ counters, headline decoder kernels (both legs of the default command)"
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_${TAG}_$i -o pmc -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-configs --no-legs > $R/$OUT/pmc_$i.log 2>&1)
  f=$(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $OUT/pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:56]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[(k, r['Counter_Name'])] += 1
for k in agg:
    if 'sdec' in k and 'reduce' not in k:
        print(k, {c: round(v / disp[(k, c)], 1) for c, v in agg[k].items()})
PY
done
echo "== HBM traffic per launch, every config (FETCH_SIZE and WRITE_SIZE in separate passes)"
rm -f $OUT/traffic.json
for spec in "C2 3" "C2 2" "C1 3" "C1 2" "C3 3" "C3 2" "C4 3" "C4 2" "C4fc 3" "C4fc 2" "C5 2" "C5 3"; do
  set -- $spec; c=$1; m=$2
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rm -rf /tmp/tr_${TAG}_${c}_${m}_$ctr && timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE $ctr --output-format csv -d /tmp/tr_${TAG}_${c}_${m}_$ctr -o pmc -- python $R/bench.py --config $c --fused $m --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-configs --no-alt --no-legs > /dev/null 2>&1)
  done
  ff=$(find /tmp/tr_${TAG}_${c}_${m}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  fw=$(find /tmp/tr_${TAG}_${c}_${m}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  [ -n "$ff" ] && [ -n "$fw" ] && python scripts/pmc_traffic.py "$c:$m" "$ff" "$fw" $OUT/traffic.json "profiles/${TAG}_traffic.json (scripts/gpu_prof_r6.sh)" | tee -a $OUT/traffic.log
done
echo "== soak: 3000 training steps in both precisions from the same seeds (the folded guide and the column-parallel tail on every step)"
timeout 900 python scripts/soak_bf16.py > $OUT/soak_bf16.txt 2>&1; tail -3 $OUT/soak_bf16.txt
ls $OUT
