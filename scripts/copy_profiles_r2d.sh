#!/bin/bash
# gpurun_out/r02d* (scripts/gpu_prof_r2d.sh) -> the tracked profiles/r02d_* set
set -e
G=gpurun_out; P=profiles
cp $G/r02d/bench.json $P/r02d_bench.json
cp $G/r02d/bench_kernel_stats.csv $P/r02d_bench_kernel_stats.csv
cp $G/r02d/pmc_summary.txt $P/r02d_pmc_summary.txt
cp $G/r02d/step_timeline_bf16.txt $P/r02d_step_timeline_bf16.txt
cp $G/r02d/traffic.json $P/r02d_traffic.json
cp $G/r02d/traffic.json $P/traffic.json
cp $G/r02d_c5/kernel_stats.csv $P/r02d_ved_b256_kernel_stats.csv
cp $G/r02d_c4/kernel_stats.csv $P/r02d_ivae64_convenc_b128_kernel_stats.csv
cp $G/r02d_c5tl/timeline.txt $P/r02d_ved_b256_step_timeline.txt
cp $G/r02d_c4tl/timeline.txt $P/r02d_ivae64_convenc_b128_step_timeline.txt
cp $G/r02d_conv_kernels.txt $P/r02d_conv_kernels.txt
cp $G/r02d_cvf/pmc_summary.txt $P/r02d_conv_fwd_pmc_summary.txt
cp $G/r02d_cvw/pmc_summary.txt $P/r02d_conv_wgrad_pmc_summary.txt
git status --short $P | head -20
