#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of a bench.py command under two values of an experiments-build
# switch: scripts/ab_kstats.sh NAME "v1 v2" [bench args]
name=$1; vals=$2; shift; shift
args=${@:---steps 100 --warmup 5 --repeats 2 --no-alt --no-configs --no-legs --no-cpu-baseline}
export PV_LIB_PATH=$PWD/pyroved_amd/libpyroved_amd_exp.so TMPDIR=/tmp
R=$PWD
for v in $vals; do
  (cd /tmp && rm -rf /tmp/ks_$v && env $name=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o t -- python $R/bench.py $args > /tmp/ks_$v.log 2>&1)
  echo "== $name=$v   $(grep BENCH-SUMMARY /tmp/ks_$v.log | cut -c1-110)"
  python - /tmp/ks_$v/t_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("   %-70s calls %5s avg %9.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
