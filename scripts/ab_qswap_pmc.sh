#!/bin/bash
# correctness + LDS counters of the fp32-class decoder launch under PV_QSWAP=0/1 (experiments build, same binary)
export TMPDIR=/tmp PV_LIB_PATH=$PWD/pyroved_amd/libpyroved_amd_exp.so
R=$PWD
for v in 0 1; do
  echo "== PV_QSWAP=$v"
  PV_QSWAP=$v python -m pytest tests/test_gpu_parity.py -q -x -k "not conv and not ved" 2>&1 | tail -1
  (cd /tmp && rm -rf /tmp/pmc_q$v && PV_QSWAP=$v timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_q$v -o pmc -- python $R/bench.py --fused 2 --steps 6 --warmup 2 --repeats 1 --no-alt --no-cpu-baseline --no-configs --no-legs > /dev/null 2>&1)
  f=$(find /tmp/pmc_q$v -name "*counter_collection.csv" | head -1)
  python - "$f" "qswap=$v" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:48]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[(k, r['Counter_Name'])] += 1
for k in agg:
    if 'fused_bf16' in k:
        d = {c: round(v / disp[(k, c)], 1) for c, v in agg[k].items()}
        print(sys.argv[2], k, d, "conflict/active = %.3f" % (d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']))
PY
done
