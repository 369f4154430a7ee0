export PV_LIB_PATH=$PWD/pyroved_amd/libpyroved_amd_exp.so
for i in 1 2; do
for a in 0 512; do
PV_FD_ABLATE=$a python bench.py --steps 100 --warmup 5 --fused 2 --no-alt --no-configs --no-legs --no-cpu-baseline 2>&1 | tail -1 | sed "s/^/ablate=$a /"
done; done
