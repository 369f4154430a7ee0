#!/bin/bash
# rocprofv3 kernel stats + SQ counters of the convolution kernels under scripts/gpu_conv_bench.py
# usage (on the box): bash scripts/gpu_conv_prof.sh <tag> <fwd|wgrad> <modes>
TAG=${1:-cv}; WHAT=${2:-fwd}; export MODES=${3:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof_$TAG -o trace -- python $R/scripts/gpu_conv_bench.py $WHAT > $R/$OUT/run.log 2>&1)
cut -d, -f1-4 /tmp/cprof_$TAG/trace_kernel_stats.csv | grep -i "conv\|Name" | cut -c1-150 | tee $OUT/kernel_stats.txt
python - /tmp/cprof_$TAG/trace_kernel_trace.csv <<'PY' | tee $OUT/kernel_times.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    if 'conv3' in r['Kernel_Name'] and 'wprep' not in r['Kernel_Name']:
        d[(r['Kernel_Name'][:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('Grid_Size_Y'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    v.sort(); print(k, "n=%d median %.1f us min %.1f" % (len(v), v[len(v) // 2], v[0]))
PY
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_IFETCH SQ_WAVE32_INSTS"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/cpmc_${TAG}_$i -o pmc -- python $R/scripts/gpu_conv_bench.py $WHAT ${PMC_B:-8} > $R/$OUT/pmc_$i.log 2>&1)
  f=$(find /tmp/cpmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $OUT/pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:48] + " g" + r.get('Grid_Size', '?')
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[(k, r['Counter_Name'])] += 1
for k in agg:
    if 'conv3' in k and 'wprep' not in k and 'finish' not in k:
        print(k, {c: round(v / disp[(k, c)], 1) for c, v in agg[k].items()})
PY
done
