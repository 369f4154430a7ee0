#!/bin/bash
# Host-side cost of one step by HIP API (medians and the per-step sum over the steady state) from a rocprofv3 --hip-trace run
CFG=${1:-C5}; FUSED=${2:-2}; export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --hip-trace --output-format csv -d /tmp/hp_$CFG -o hp -- python $GRAFT_REPO_ROOT/scripts/host_bound.py $CFG $FUSED 100 2>/dev/null | grep "host enqueue"
f=$(find /tmp/hp_$CFG -name "*hip_api_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows); rows = rows[n // 2:]            # the steady state: the second half of the calls
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
by = {}
for r in rows:
    by.setdefault(r["Function"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in by.values())
print("window %.1f ms, inside HIP calls %.1f ms (%.0f %%)" % ((t1 - t0) / 1e6, tot / 1e6, 100.0 * tot / (t1 - t0)))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("  %-28s calls %6d  median %7.2f us  mean %7.2f us  total %8.2f ms" % (k, len(v), st.median(v) / 1e3, st.mean(v) / 1e3, sum(v) / 1e6))
PY
