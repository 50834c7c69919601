#!/bin/bash
# run bench.py once per variant library: tools/ab.sh TAG NAME1 NAME2 ...   (results: gpurun_out/ab_TAG_NAME.{json,err})
TAG=$1; shift
mkdir -p gpurun_out
for n in "$@"; do
  if [ "$n" == "base" ]; then L=""; else L="$PWD/_variants/libyololite_hip_$n.so"; fi
  YOLOLITE_HIP_LIB=$L python bench.py --steps 30 --warmup 5 --no-cpu-baseline --layers > gpurun_out/ab_${TAG}_$n.json 2> gpurun_out/ab_${TAG}_$n.err
  python - <<PY
import json; d=json.load(open("gpurun_out/ab_${TAG}_$n.json")); print("$n", d["value"], d["network"]["forward_ms_sum_of_layers"])
PY
done
