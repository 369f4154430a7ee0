for i in 1 2; do
for l in pyroved_amd/libpyroved_amd.so pyroved_amd/variants/lib_gb4.so pyroved_amd/variants/lib_dgd2.so pyroved_amd/variants/lib_gb4dgd2.so pyroved_amd/variants/lib_gb1.so; do
  PV_LIB_PATH=$PWD/$l python bench.py --steps 100 --warmup 5 --fused 2 --no-alt --no-configs --no-legs --no-cpu-baseline 2>&1 | tail -1 | sed "s|^|$(basename $l) |" | cut -c1-150
done; done
