#!/bin/bash
# Round-2 profiling run on the GPU box: bench line, rocprofv3 kernel stats, step timeline, PMC passes of the decoder
# kernel (SQ counters; FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes), and
# profiles/traffic.json (HBM bytes per launch of the dominant kernel, FETCH doubled per the guide's gfx950 correction).
# Usage (from the repo root, on the box): bash scripts/gpu_prof_r2.sh <tag>     -> gpurun_out/<tag>/
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
echo "== bench"; timeout 900 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; python scripts/benchline.py < $OUT/bench.json
echo "== rocprof kernel stats (the default bench command, headline legs only)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python $R/bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-configs > $R/$OUT/rocprof.log 2>&1)
cp /tmp/prof_$TAG/trace_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
echo "== timeline"
BENCH_ARGS="--no-alt --no-configs" bash scripts/gpu_timeline.sh ${TAG}_tl > /dev/null 2>&1; cp gpurun_out/${TAG}_tl/timeline.txt $OUT/step_timeline_bf16.txt; cat $OUT/step_timeline_bf16.txt
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_${TAG}_$i -o pmc -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-configs > $R/$OUT/pmc_$i.log 2>&1)
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
python - $OUT/pmc_summary.txt <<'PY'
import ast, json, re, sys
vals = {}
for line in open(sys.argv[1]):
    m = re.match(r"(.*?) (\{.*\})$", line.strip())
    if not m: continue
    name, d = m.group(1), ast.literal_eval(m.group(2))
    key = "C2:3" if "w8" in name else ("C2:2" if "true>" in name.replace(" ", "") and "bf16" in name else None)
    if key is None: continue
    vals.setdefault(key, {}).update(d)
out = {}
for key, d in vals.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # rocprofv3 reports KiB; FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports half of a wide streaming read)
        out[key] = {"bytes": int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024), "source": "profiles/%s_pmc_summary.txt" % sys.argv[1].split("/")[-2],
                    "fetch_kib": d["FETCH_SIZE"], "write_kib": d["WRITE_SIZE"]}
json.dump(out, open(sys.argv[1].replace("pmc_summary.txt", "traffic.json"), "w"), indent=1)
print(out)
PY
ls $OUT
