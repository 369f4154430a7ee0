#!/bin/bash
# rocprofv3 kernel stats of one config's SVI step (default C5, fp32-class leg): bash scripts/gpu_ved_prof.sh <tag> [C5|C4] [extra bench args]
TAG=${1:-vp}; CFG=${2:-C5}; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vprof_$TAG -o trace -- python $R/bench.py --config $CFG --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline "$@" > $R/$OUT/run.log 2>&1)
cp /tmp/vprof_$TAG/trace_kernel_stats.csv $OUT/kernel_stats.csv
python - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print("%-70s calls %5s  total %8.1f us  avg %7.1f us  %5.1f%%" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
tail -1 $OUT/run.log | cut -c1-200
