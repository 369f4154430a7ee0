#!/bin/bash
# Step period by position in a timed region (bench.py --steps K): is a short region slower per step, and where?
K=${1:-20}; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && rm -rf /tmp/rp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp -o rp -- python $R/bench.py --steps $K --warmup 5 --repeats 3 --no-cpu-baseline --no-configs --no-legs --no-alt > /dev/null 2>&1
f=$(find /tmp/rp -name "*kernel_trace.csv" | head -1)
python - "$f" $K <<'PY'
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1]))); K = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dec = [r for r in rows if "pv_sdec_w8_kernel" in r["Kernel_Name"]]
starts = [int(r["Start_Timestamp"]) for r in dec]; durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in dec]
# regions: a gap of more than 3 step periods between decoder launches starts a new one
per = [starts[i + 1] - starts[i] for i in range(len(starts) - 1)]
med = st.median(per)
regs, cur = [], [0]
for i, p in enumerate(per):
    if p > 3 * med: regs.append(cur); cur = [i + 1]
    else: cur.append(i + 1)
regs.append(cur)
regs = [r for r in regs if len(r) == K]
print("%d regions of %d steps; median period %.2f us" % (len(regs), K, med / 1e3))
for pos in range(K):
    ps = [per[r[pos]] for r in regs if pos + 1 < K]
    ds = [durs[r[pos]] for r in regs]
    print("  step %2d: decoder launch %.2f us%s" % (pos, st.mean(ds) / 1e3, ("  period %.2f us" % (st.mean(ps) / 1e3)) if ps else ""))
PY
