#!/bin/bash
# developer run on the GPU box: masks A/B (see tools/masks_ab.py) + config-4 bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/mab; mkdir -p $O
for cfg in "2048:4096" "2048:8192" "2048:16384" "1024:8192" "3072:8192"; do
  echo "blocks fill:box=$cfg"; YL_MI_BLOCKS=$cfg timeout -k 5 300 python tools/masks_ab.py 2>&1 | grep "launch ms"
done
