#!/bin/bash
# developer A/B on the GPU box:  tools/run_ab.sh SCRIPT.py [old-lib]   (default old lib: _variants/libyololite_hip_r02.so)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OLD=${2:-_variants/libyololite_hip_r02.so}
O=gpurun_out/ab; mkdir -p $O
YOLOLITE_HIP_LIB=$GRAFT_REPO_ROOT/$OLD timeout -k 5 400 python $1 2>&1 | grep -v amdgpu.ids > $O/old.txt
timeout -k 5 400 python $1 2>&1 | grep -v amdgpu.ids > $O/new.txt
diff <(grep -v "^lib\| ms" $O/old.txt) <(grep -v "^lib\| ms" $O/new.txt) && echo SAME
echo "--- old"; cat $O/old.txt; echo "--- new"; cat $O/new.txt
