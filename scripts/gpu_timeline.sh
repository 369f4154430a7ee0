#!/bin/bash
# Kernel timeline of one SVI step (start offset, duration, gap to the previous kernel) from a rocprofv3 kernel trace.
TAG=${1:-tl}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o tl -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline ${BENCH_ARGS} > $R/$OUT/run.log 2>&1)
f=$(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts at each enc_fwd kernel; take the 3rd-from-last complete step
starts = [i for i, n in enumerate(names) if "enc_l1" in n or "pv_enc_kernel" in n or "pv_guide_img_kernel" in n]
if len(starts) < 3:      # (round 5: the decoder launch hosts the guide — a step starts at that launch)
    starts = [i for i, n in enumerate(names) if "pv_sdec_w8_kernel<true, " in n and ("true>" in n or ", 1>" in n or ", 2>" in n or ", 3>" in n)]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None
tot_k = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else s - prev_end
    print("%9.2f us  dur %8.2f  gap %7.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r["Kernel_Name"][:70]))
    prev_end = e; tot_k += e - s
print("step span %.2f us, kernel time %.2f us, next step starts at %.2f us" % ((prev_end - t0) / 1e3, tot_k / 1e3, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
