#!/bin/bash
# round-2 (d) artefacts: headline profile + conv-path profiles (C4 / C5 kernel stats and step timelines, conv kernel bench)
set -x
bash scripts/gpu_prof_r2.sh r02d > gpurun_out/r02d_prof.log 2>&1
bash scripts/gpu_ved_prof.sh r02d_c5 C5 > gpurun_out/r02d_c5_stats.txt 2>&1
bash scripts/gpu_ved_prof.sh r02d_c4 C4 > gpurun_out/r02d_c4_stats.txt 2>&1
bash scripts/gpu_timeline_cfg.sh r02d_c5tl C5 > gpurun_out/r02d_c5_timeline.txt 2>&1
bash scripts/gpu_timeline_cfg.sh r02d_c4tl C4 > gpurun_out/r02d_c4_timeline.txt 2>&1
MODES=0,1,2,3,4 python scripts/gpu_conv_bench.py all > gpurun_out/r02d_conv_kernels.txt 2>&1
PMC_B=256 bash scripts/gpu_conv_prof.sh r02d_cvf fwd 4 > gpurun_out/r02d_conv_fwd_pmc.txt 2>&1
PMC_B=256 bash scripts/gpu_conv_prof.sh r02d_cvw wgrad 4 > gpurun_out/r02d_conv_wgrad_pmc.txt 2>&1
ls gpurun_out
