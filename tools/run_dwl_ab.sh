#!/bin/bash
# yl_conv_dwl_kernel A/B on the MI355X box: bitwise test, then per-layer eager times + pipelined step with the old kernel
# ("dev_select" 64 = yl_conv_dws_kernel, 16384 = yl_conv_dwk_kernel) and the new one (0).  tools/run_dwl_ab.sh [model] [batch] [extra bench flags]
M=${1:-edge_m}; B=${2:-32}; X=${3:-}
python -m pytest tests/test_gpu_parity.py -q -x -k "window_in_lds" 2>&1 | tail -5
for dv in ${DVS:-64 16384 0}; do
  echo "== dev_select $dv"
  python bench.py --model $M --batch $B --steps 15 --warmup 3 --no-cpu-baseline --layers $X --opt dev_select=$dv > /tmp/ab_$dv.json 2> /tmp/ab_$dv.txt
  grep -E "dw3" /tmp/ab_$dv.txt | grep -E "cin (244|328)" | cut -c1-140
  python -c "
import json,sys
d=json.loads(open('/tmp/ab_$dv.json').read().strip().splitlines()[-1]); print('images/s', d['value'], 'ms/step', d['ms_per_step'], 'one-in-flight', (d.get('one_batch_in_flight') or {}).get('value'))"
done
