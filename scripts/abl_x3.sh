for round in 1 2; do
for v in base x3old; do
  if [ $v = base ]; then L=""; else L="PV_LIB_PATH=pyroved_amd/variants/lib_$v.so"; fi
  for B in 256 2048; do
  echo -n "$v B=$B: "
  env $L timeout 300 python bench.py --config C2 --fused 2 --batch $B --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>/dev/null | python scripts/benchline.py
  done
done
done
