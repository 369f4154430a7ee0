timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or h2_kernel or w8_kernel or full_size_properties or one_call" 2>&1 | tail -2
for round in 1 2 3; do
  echo -n "C2 bf16: "; timeout 300 python bench.py --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>/dev/null | python scripts/benchline.py
  echo -n "C2 fp32-class: "; timeout 300 python bench.py --fused 2 --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-legs --no-alt --no-configs 2>/dev/null | python scripts/benchline.py
done
