#!/bin/bash
# A/B of a variant library against the shipped one on bench.py lines: scripts/ab_lib2.sh <variant> "<bench args>" [rounds]
V=$PWD/pyroved_amd/variants/lib_$1.so; args=$2; n=${3:-3}
for i in $(seq $n); do
  for l in $PWD/pyroved_amd/libpyroved_amd.so $V; do
    PV_LIB_PATH=$l python bench.py $args 2>&1 | tail -1 | sed "s|^|$(basename $l) |" | cut -c1-200
  done
done
