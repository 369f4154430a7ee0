#!/bin/bash
# builds pyroved_amd/variants/lib_<name>.so from the bf16 kernel with extra -D flags:  mkvariant.sh name -DFB_DGD=2 ...
set -e
cd "$(dirname "$0")/../pyroved_amd/csrc"
name=$1; shift
mkdir -p ../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c pv_sdec_fused_bf16.hip -o /tmp/var_$name.o "$@"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC pv_gemm.o pv_wgrad.o pv_elementwise.o pv_encoder.o pv_plan.o pv_sdec_fused.o pv_conv.o pv_conv_direct.o pv_ved.o pv_ss.o /tmp/var_$name.o -o ../variants/lib_$name.so
echo built lib_$name.so
