#!/bin/bash
# The conv path's backward forms under EQUAL decisions: error of every conv-stack gradient against the float64
# oracle run with the HIP forward's own max-pool winners / leaky-ReLU signs, and the step time, for
#   dg = input gradient: 5 = dL/dy ONE fp16 piece x two-piece weights (2 products), 4 = both split (3 products)
#   wg = weight gradient: 1 = one piece per operand (1 product), 4 = both split (3 products)
# (experiments build: PV_CONV_DG / PV_CONV_WG).  Writes gpurun_out/conv_bwd_forms.txt.
cd "$(dirname "$0")/.."
export PV_LIB_PATH=$PWD/pyroved_amd/libpyroved_amd_exp.so
out=gpurun_out/conv_bwd_forms.txt
: > $out
for form in "5 1" "4 1" "5 4" "4 4"; do
  set -- $form
  export PV_CONV_DG=$1 PV_CONV_WG=$2
  echo "==== dg=$1 wg=$2" >> $out
  rm -f gpurun_out/grad_margin_masked.txt
  python -m pytest tests/test_gpu_parity.py -q -k "(full_size_c4 and 2-) or (full_size_c5 and fp32-)" 2>&1 | tail -1 >> $out
  grep -E "^C[45]|layers.0.weight|layers.3.weight|layers.10.weight|worst" gpurun_out/grad_margin_masked.txt >> $out
  for c in C4 C5; do
    python bench.py --config $c --steps 20 --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$c ms_per_step', round(d['ms_per_step'], 4))" >> $out
  done
done
cat $out
