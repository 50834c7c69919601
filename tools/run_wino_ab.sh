#!/bin/bash
# Winograd kernel A/B on the MI355X box: the bitwise test, then per-layer eager times + pipelined step for every item shape
# ("dev_select": 2048 = first form, 0 auto, 4096 (4,4), 8192 (2,7), 12288 two m-tiles).  tools/run_wino_ab.sh [model] [batch]
M=${1:-yololite_m}; B=${2:-32}; X=${3:-}
python -m pytest tests/test_gpu_parity.py -q -x -k "winograd" 2>&1 | tail -5
for dv in ${DVS:-2048 0 4096 8192 12288}; do
  echo "== dev_select $dv"
  python bench.py --model $M --batch $B --steps 15 --warmup 3 --no-cpu-baseline --layers $X --opt dev_select=$dv > /tmp/ab_$dv.json 2> /tmp/ab_$dv.txt
  grep -E "smooth|proto" /tmp/ab_$dv.txt | cut -c1-140
  python -c "
import json,sys
d=json.loads(open('/tmp/ab_$dv.json').read().strip().splitlines()[-1]); print('images/s', d['value'], 'ms/step', d['ms_per_step'], 'one-in-flight', (d.get('one_batch_in_flight') or {}).get('value'))"
done
