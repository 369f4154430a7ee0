#!/bin/bash
# builds pyroved_amd/variants/lib_prev.so: the working tree's objects with the named sources taken from a git revision instead
#   scripts/mkvariant_prev.sh HEAD pv_sdec_fused_w8.hip pv_sdec_fused_bf16.hip      (A/B against the last commit: scripts/ab_lib2.sh prev ...)
set -e
cd "$(dirname "$0")/../pyroved_amd/csrc"
rev=$1; shift
mkdir -p ../variants /tmp/pv_prev
skip=""
for src in "$@"; do
  git show $rev:pyroved_amd/csrc/$src > _prev_$src
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c _prev_$src -o /tmp/pv_prev/${src%.hip}.o
  rm -f _prev_$src
  skip="$skip|${src%.hip}.o"
done
objs=$(ls *.o | grep -vE "^(${skip#|})$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $(for s in "$@"; do echo /tmp/pv_prev/${s%.hip}.o; done) -o ../variants/lib_prev.so
echo built lib_prev.so
