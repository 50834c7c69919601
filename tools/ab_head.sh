#!/bin/bash
# Same-box A/B of the working tree against HEAD:  tools/ab_head.sh [bench args...]
#   builds _variants/libyololite_hip_new.so (working tree) and _variants/libyololite_hip_head.so (git stash),
#   then prints the gpurun command that alternates the two on ONE box (box-to-box noise is +-1 %).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
mkdir -p _variants
python yololite-official-repo_amd/csrc/build.py > /dev/null 2>&1
cp yololite-official-repo_amd/libyololite_hip.so _variants/libyololite_hip_new.so
git stash -q
python yololite-official-repo_amd/csrc/build.py > /dev/null 2>&1
cp yololite-official-repo_amd/libyololite_hip.so _variants/libyololite_hip_head.so
git stash pop -q
python yololite-official-repo_amd/csrc/build.py > /dev/null 2>&1
echo "for i in 1 2 3 4; do for v in head new; do echo \"\$v \$(YOLOLITE_HIP_LIB=_variants/libyololite_hip_\$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline $* | cut -c30-50)\"; done; done"
