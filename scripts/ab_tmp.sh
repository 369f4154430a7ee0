for v in base skew16 skew48 skew96; do
  if [ $v = base ]; then L=""; else L="PV_LIB_PATH=pyroved_amd/variants/lib_$v.so"; fi
  echo "== $v"
  env $L MODES=4,7 timeout 200 python scripts/gpu_conv_bench.py fwd 256 2>&1 | grep -v amdgpu
done
