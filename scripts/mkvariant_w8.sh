#!/bin/bash
# builds pyroved_amd/variants/lib_<name>.so with the 8-wave decoder kernel compiled with extra -D flags:
#   scripts/mkvariant_w8.sh trace -DW8_TRACE        (run with PV_LIB_PATH=pyroved_amd/variants/lib_trace.so)
set -e
cd "$(dirname "$0")/../pyroved_amd/csrc"
name=$1; shift
mkdir -p ../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c pv_sdec_fused_w8.hip -o /tmp/varw8_$name.o "$@"
objs=$(ls *.o | grep -v pv_sdec_fused_w8.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/varw8_$name.o -o ../variants/lib_$name.so
echo built lib_$name.so
