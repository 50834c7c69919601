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


// clamp to [lo,hi] in ONE VALU op per element (v_med3_f32).  fp32 MFMA and fp32 VALU share the SIMD's FMA
// lanes on gfx950 (same 64 FLOP/clk/SIMD peak; measured: removing VALU work shortens MFMA-bound kernels 1:1),
// so epilogue instruction count is kernel time.  lo = -inf / hi = +inf give the one-sided / identity cases.
__device__ __forceinline__ f32x4 yl_clamp4(f32x4 v, float lo, float hi) {
  f32x4 r;
  r.x = __builtin_amdgcn_fmed3f(v.x, lo, hi); r.y = __builtin_amdgcn_fmed3f(v.y, lo, hi);
  r.z = __builtin_amdgcn_fmed3f(v.z, lo, hi); r.w = __builtin_amdgcn_fmed3f(v.w, lo, hi);
  return r;
}
