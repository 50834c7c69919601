#!/bin/bash
# A/B helper: build a variant of libyololite_hip.so with extra -D flags for ONE translation unit.
#   tools/build_variant.sh NAME UNIT.hip -DFOO=1 ...   ->  _variants/libyololite_hip_NAME.so
# Run with  YOLOLITE_HIP_LIB=_variants/libyololite_hip_NAME.so python bench.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/yololite-official-repo_amd/csrc
NAME=$1; UNIT=$2; shift 2
mkdir -p $ROOT/_variants/obj_$NAME
EXTRA=""
case $UNIT in yl_post.hip|yl_pre.hip|yl_eval.hip|yl_track.hip|yl_ops.hip) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value $EXTRA "$@" \
  -c $CS/$UNIT -o $ROOT/_variants/obj_$NAME/${UNIT%.hip}.o
OBJS=""
for u in yl_api yl_conv yl_stemblock yl_convc yl_dpp yl_se yl_ops yl_conv_bf16 yl_stemblock_bf16 yl_convc_bf16 yl_conv_f16 yl_stemblock_f16 yl_convc_f16 yl_conv_f16s yl_stemblock_f16s yl_convc_f16s yl_post yl_pre yl_eval yl_track; do
  if [ "$u.hip" == "$UNIT" ]; then OBJS="$OBJS $ROOT/_variants/obj_$NAME/$u.o"; else OBJS="$OBJS $CS/_obj/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/_variants/libyololite_hip_$NAME.so $OBJS
echo $ROOT/_variants/libyololite_hip_$NAME.so
