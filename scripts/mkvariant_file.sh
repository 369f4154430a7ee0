#!/bin/bash
# builds pyroved_amd/variants/lib_<name>.so with ONE source compiled with extra -D flags:
#   scripts/mkvariant_file.sh lbtrace pv_elementwise.hip -DLB_TRACE     (run with PV_LIB_PATH=pyroved_amd/variants/lib_lbtrace.so)
set -e
cd "$(dirname "$0")/../pyroved_amd/csrc"
name=$1; src=$2; shift; shift
mkdir -p ../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c $src -o /tmp/var_$name.o "$@"
objs=$(ls *.o | grep -v "${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/var_$name.o -o ../variants/lib_$name.so
echo built lib_$name.so
