#!/bin/bash
# packed bf16 gradient records of the throughput kernel (default) against the fp32 form (PV_FD_ABLATE=1024, experiments build)
export PV_LIB_PATH=$PWD/pyroved_amd/libpyroved_amd_exp.so
for i in 1 2 3; do
for a in 0 1024; do
PV_FD_ABLATE=$a python bench.py --steps 200 --warmup 5 --no-alt --no-configs --no-legs --no-cpu-baseline 2>&1 | tail -1 | sed "s/^/records_fp32=$a /"
done; done
PV_FD_ABLATE=0 python bench.py --config C3 --fused 3 --steps 20 --warmup 5 --no-alt --no-configs --no-legs --no-cpu-baseline 2>&1 | tail -1 | sed "s/^/C3 packed /"
PV_FD_ABLATE=1024 python bench.py --config C3 --fused 3 --steps 20 --warmup 5 --no-alt --no-configs --no-legs --no-cpu-baseline 2>&1 | tail -1 | sed "s/^/C3 fp32 /"
