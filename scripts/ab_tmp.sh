timeout 900 python -m pytest tests -x -q -m gpu -k "conv or ved or VED or c4 or c5 or two_stream or convnet or c1" 2>&1 | tail -2
for round in 1 2; do
for v in oldc1 base; do
  if [ $v = base ]; then L=""; else L="PV_LIB_PATH=pyroved_amd/variants/lib_$v.so"; fi
  for c in C5 C4; do for f in 2 3; do echo -n "$v $c fused=$f: "; env $L timeout 300 python bench.py --config $c --fused $f --steps 40 --warmup 10 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>/dev/null | python scripts/benchline.py; done; done
done
done
