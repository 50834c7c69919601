#!/bin/bash
# per-layer table (first N layers) for variant libraries: tools/ab_layers.sh N NAME...
N=$1; shift
for n in "$@"; do
  if [ "$n" == "base" ]; then L=""; else L="$PWD/_variants/libyololite_hip_$n.so"; fi
  echo "== $n"; YOLOLITE_HIP_LIB=$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --layers 2>&1 >/dev/null | grep -v amdgpu.ids | head -$N
done
