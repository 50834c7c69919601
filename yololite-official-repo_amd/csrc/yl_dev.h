// Small device helpers shared by the conv translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/yololite_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float yl_act1(float v, int act) {
  switch (act) {
    case YL_ACT_RELU: return fmaxf(v, 0.0f);
    case YL_ACT_RELU6: return fminf(fmaxf(v, 0.0f), 6.0f);
    case YL_ACT_SILU: return v / (1.0f + expf(-v));
    default: return v;
  }
}
__device__ __forceinline__ f32x4 yl_act4(f32x4 v, int act) {
  f32x4 r;
  r.x = yl_act1(v.x, act); r.y = yl_act1(v.y, act); r.z = yl_act1(v.z, act); r.w = yl_act1(v.w, act);
  return r;
}
__device__ __forceinline__ f32x4 yl_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

