#!/bin/bash
# alternating runs of bench.py under several libraries: scripts/ab_libs.sh "<lib names in pyroved_amd/variants, or 'main'>" "<bench args>" [rounds]
libs=$1; args=$2; n=${3:-3}
for i in $(seq $n); do
  for l in $libs; do
    p=$PWD/pyroved_amd/variants/lib_$l.so; [ "$l" = main ] && p=$PWD/pyroved_amd/libpyroved_amd.so
    PV_LIB_PATH=$p python bench.py $args 2>&1 | tail -1 | sed "s|^|$l |" | cut -c1-190
  done
done
