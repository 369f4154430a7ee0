#!/bin/bash
# Profiling run on the GPU box: correctness subset, bench, phase ablations of the fused kernel,
# rocprofv3 kernel trace and PMC passes.  Usage: scripts/gpu_prof.sh <tag> [ablation-list]
TAG=${1:-prof}; ABL=${2:-"0 1 2 4 8 15"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest"; timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
echo "== bench"; timeout 600 python bench.py --steps 100 --warmup 20 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('value %.0f img/s  ms/step %.4f  kernel_ms %.4f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
echo "== ablations (PV_FD_ABLATE: 1 wgrad exch, 2 coord exch, 4 dgrad, 8 dwo)"
for A in $ABL; do
  PV_FD_ABLATE=$A timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('ablate=$A ms/step %.4f kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" | tee -a $OUT/ablate.log
done
echo "== rocprof trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1)
cp /tmp/prof_$TAG/trace_kernel_stats.csv $OUT/ 2>/dev/null
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_${TAG}_$i -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/$OUT/pmc_$i.log 2>&1)
  f=$(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'EOF' | tee -a $OUT/pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:48]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
disp = collections.Counter()
for r in rows:
    disp[(r['Kernel_Name'][:48], r['Counter_Name'])] += 1
for k in agg:
    if 'fused' in k and 'reduce' not in k:
        print(k, {c: round(v / disp[(k, c)], 1) for c, v in agg[k].items()})
EOF
done
ls $OUT
