#!/bin/bash
# Ablation variants of yl_conv_dws_kernel (results wrong, timing only):  for a in 1 2 3 7; do tools/build_variant.sh dwsabl$a yl_convc.hip -DYL_DWS_ABL=$a; done
#   1 = one tap-weight LDS read per tap row instead of DK, 2 = one tap read per row, 3 = both, 7 = and one fma per row
for a in 0 1 2 3 7; do
  L=_variants/libyololite_hip_dwsabl$a.so; [ $a == 0 ] && L=yololite-official-repo_amd/libyololite_hip.so
  echo "== ablation $a"
  YL_BENCH_ALLOW_EMPTY=1 YOLOLITE_HIP_LIB=$L python bench.py --model yololite_m --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --layers --min-seconds 0 --in-flight 1 2> /tmp/abl.txt > /dev/null
  grep -E "blocks.4.[01].conv_pwl|blocks.5.[01].conv_pwl" /tmp/abl.txt | cut -c1-120
done
