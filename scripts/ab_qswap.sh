#!/bin/bash
# A/B of the q-swapped weight-image column order (pv_fb_layout.h) on the fp32-class step: same binary, PV_QSWAP=0/1
bash scripts/ab_env.sh PV_QSWAP "0 1" --steps 200 --warmup 5 --fused 2 --no-alt --no-configs --no-legs --no-cpu-baseline
bash scripts/ab_kstats.sh PV_QSWAP "0 1" --steps 100 --warmup 5 --repeats 2 --fused 2 --no-alt --no-configs --no-legs --no-cpu-baseline
