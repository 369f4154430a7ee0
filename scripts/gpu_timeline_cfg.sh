#!/bin/bash
# Kernel timeline of one SVI step of a bench config (a step = the kernels between two pv_adam launches, or two pv_wgrad_small launches
# when Adam rides in that launch), last leg run:
#   bash scripts/gpu_timeline_cfg.sh <tag> <C4|C5> [extra bench args]
TAG=${1:-tlc}; CFG=${2:-C5}; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlc_$TAG -o tl -- python $R/bench.py --config $CFG --steps 8 --warmup 3 --repeats 1 --no-cpu-baseline "$@" > $R/$OUT/run.log 2>&1)
f=$(find /tmp/tlc_$TAG -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, n in enumerate(names) if "pv_adam" in n]
if len(ends) < 3:                                     # (conv-encoder iVAE, round 4: Adam rides in the step's last weight-gradient launch)
    ends = [i for i, n in enumerate(names) if "pv_wgrad_small" in n]
a, b = ends[-3] + 1, ends[-2] + 1
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None
tot_k = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else s - prev_end
    g = r.get("Grid_Size", "?"); w = r.get("Workgroup_Size", "?")
    print("%9.2f us  dur %8.2f  gap %6.2f  %-58s grid %s/%s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r["Kernel_Name"][:58], g, w))
    prev_end = e; tot_k += e - s
print("step span %.2f us, kernel time %.2f us, %d kernels" % ((prev_end - t0) / 1e3, tot_k / 1e3, b - a))
PY
