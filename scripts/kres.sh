#!/bin/bash
# scripts/kres.sh <file.hip> [name filter] — rebuild the library and print register / scratch use of one source's kernels
set -e
cd /root/repo/pyroved_amd/csrc
make -j8 2>&1 | grep -E "error|Error" | head || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/kres.o -save-temps=obj 2>&1 | grep -E "error" | head || true
grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|name:)|\.vgpr_spill" /tmp/$(basename "$1" .hip)-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - - | grep "${2:-.}" | sed 's/ \+/ /g'
