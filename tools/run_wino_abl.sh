#!/bin/bash
# Ablation variants of yl_conv_wino2_kernel (results wrong, timing only): which of raw copies (1), U loads (2), transform (4)
# costs what.  Build here: for a in 1 2 4 7; do tools/build_variant.sh wabl$a yl_convc.hip -DYL_WINO_ABL=$a; done
for a in 0 1 2 7; do
  L=_variants/libyololite_hip_wabl$a.so; [ $a == 0 ] && L=yololite-official-repo_amd/libyololite_hip.so
  echo "== ablation $a"
  YOLOLITE_HIP_LIB=$L python bench.py --model yololite_m --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --layers --min-seconds 0 --in-flight 1 2> /tmp/abl.txt > /dev/null
  grep -E "smooth[34]\.0" /tmp/abl.txt | cut -c1-120
done
