#!/bin/bash
# Gradient-record formats (pv_sdec_fused.h PV_REC_*): lane-native (default: fp32 for the 4-wave kernels, packed bf16 pairs for the
# 8-wave throughput kernel) against the row-major fp32 form of rounds 1-5 (PV_FD_ABLATE=1024, experiments build); same box.
export PV_LIB_PATH=$PWD/pyroved_amd/libpyroved_amd_exp.so
for i in 1 2 3; do
for a in 0 1024; do
PV_FD_ABLATE=$a python bench.py --steps 200 --warmup 5 --no-alt --no-configs --no-legs --no-cpu-baseline 2>&1 | tail -1 | cut -c1-150 | sed "s/^/bf16 rowmajor_fp32=$a /"
PV_FD_ABLATE=$a python bench.py --steps 100 --warmup 5 --fused 2 --no-alt --no-configs --no-legs --no-cpu-baseline 2>&1 | tail -1 | cut -c1-150 | sed "s/^/fp32class rowmajor_fp32=$a /"
done; done
for a in 0 1024; do
PV_FD_ABLATE=$a python bench.py --config C3 --steps 20 --warmup 5 --no-alt --no-configs --no-legs --no-cpu-baseline 2>&1 | tail -1 | cut -c1-150 | sed "s/^/C3 fp32class rowmajor_fp32=$a /"
done
