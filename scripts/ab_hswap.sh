#!/bin/bash
# the image layout with the 8-byte halves swapped on alternate 4-row groups (-DW8_HSWAP=1, variants/lib_hswap.so:
# conflict-free transposing reads, two ds_read_b64 per forward operand) against the shipped layout — correctness, step time,
# kernel time and SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the headline launch.
export TMPDIR=/tmp
R=$PWD
V=$R/pyroved_amd/variants/lib_${1:-hswap}.so
echo "== correctness of the variant (folded guide + decoder vs the oracle)"
PV_LIB_PATH=$V python -m pytest tests/test_gpu_parity.py -q -k "guide_folded and rt_b256" 2>&1 | tail -1
PV_LIB_PATH=$V python -m pytest tests/test_gpu_parity.py -q -k "bf16_mode_steps and rt_b256" 2>&1 | tail -1
echo "== step time (bench.py C2 bf16, --steps 200), alternating"
for i in 1 2 3; do
  for l in $R/pyroved_amd/libpyroved_amd.so $V; do
    PV_LIB_PATH=$l python bench.py --steps 200 --warmup 5 --no-alt --no-configs --no-legs --no-cpu-baseline 2>&1 | tail -1 | sed "s|^|$(basename $l) |" | cut -c1-190
  done
done
echo "== LDS counters of the hosting launch"
for l in $R/pyroved_amd/libpyroved_amd.so $V; do
  n=$(basename $l .so)
  (cd /tmp && rm -rf /tmp/pmc_$n && PV_LIB_PATH=$l timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_$n -o pmc -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-alt --no-cpu-baseline --no-configs --no-legs > /dev/null 2>&1)
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python - "$f" "$n" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:48]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[(k, r['Counter_Name'])] += 1
for k in agg:
    if 'sdec_w8' in k:
        d = {c: round(v / disp[(k, c)], 1) for c, v in agg[k].items()}
        print(sys.argv[2], k, d, "conflict/active = %.3f" % (d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']))
PY
done
