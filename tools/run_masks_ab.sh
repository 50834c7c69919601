#!/bin/bash
# developer run on the GPU box: masks A/B (see tools/masks_ab.py) + kernel duration from rocprofv3 + config-4 bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/mab; mkdir -p $O
if [ -f _variants/libyololite_hip_r02.so ] && [ "$1" == "ab" ]; then
  YOLOLITE_HIP_LIB=$GRAFT_REPO_ROOT/_variants/libyololite_hip_r02.so timeout -k 5 400 python tools/masks_ab.py --time 0 > $O/old.txt 2>&1
fi
timeout -k 5 400 python tools/masks_ab.py > $O/new.txt 2>&1
[ -f $O/old.txt ] && diff <(grep -v "^lib\|call ms" $O/old.txt) <(grep -v "^lib\|call ms" $O/new.txt) && echo SAME
cat $O/new.txt | grep -v amdgpu.ids
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/prof -o new --output-format csv -- python tools/masks_ab.py --time 10 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -1 $f; grep -i "mask" $f
timeout -k 5 400 python bench.py --model edge_m --seg 1 --batch 32 --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
rm -rf $O/prof
