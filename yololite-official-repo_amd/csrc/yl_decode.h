// Anchor-free decode arithmetic shared by the post-processing kernels (yl_post.hip, compiled with
// -ffp-contract=off) and the fused decode epilogue of the head-output conv (yl_conv.hip, compiled with the
// default contraction): every function pins `fp contract(off)` itself, so both users evaluate the reference
// expressions (scripts/helpers/utils_ms.py:71-106) with one rounding per operation, bit-identically.
#pragma once
#include "yl_internal.h"
#include <math.h>

__device__ __forceinline__ float yl_sigmoid(float x) {
#pragma clang fp contract(off)
  return 1.0f / (1.0f + expf(-x));
}
__device__ __forceinline__ float yl_softplus(float x) {
#pragma clang fp contract(off)
  return x > 20.0f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float yl_clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__device__ __forceinline__ int yl_level_of(const YlLevels& lv, int n) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < YL_MAX_LEVELS; ++i)
    if (i < lv.L && n >= lv.off[i]) l = i;
  return l;
}

// centre / size of one candidate: grid cell (gx, gy), level stride st
__device__ __forceinline__ void yl_decode_cell(float gx, float gy, float st, float tx, float ty, float tw, float th,
                                               int center_mode, int wh_mode, float& px, float& py, float& pw,
                                               float& ph) {
#pragma clang fp contract(off)
  const float sx = yl_sigmoid(tx), sy = yl_sigmoid(ty);
  if (center_mode == YL_CENTER_V8) {
    px = ((sx * 2.0f - 0.5f) + gx) * st;
    py = ((sy * 2.0f - 0.5f) + gy) * st;
  } else {
    px = (sx + gx) * st;
    py = (sy + gy) * st;
  }
  if (wh_mode == YL_WH_SOFTPLUS) {
    pw = yl_softplus(tw) * st;
    ph = yl_softplus(th) * st;
  } else if (wh_mode == YL_WH_V8) {
    const float a = yl_sigmoid(tw) * 2.0f, b = yl_sigmoid(th) * 2.0f;
    pw = (a * a) * st;
    ph = (b * b) * st;
  } else {
    pw = expf(yl_clampf(tw, -4.0f, 4.0f)) * st;
    ph = expf(yl_clampf(th, -4.0f, 4.0f)) * st;
  }
}

__device__ __forceinline__ void yl_decode_box(const YlLevels& lv, int l, int r, float tx, float ty, float tw,
                                              float th, int center_mode, int wh_mode, float& px, float& py,
                                              float& pw, float& ph) {
  const int S = lv.S[l];
  const int cell = r % (S * S);
  yl_decode_cell((float)(cell % S), (float)(cell / S), lv.stride[l], tx, ty, tw, th, center_mode, wh_mode, px, py, pw,
                 ph);
}

// corners, clamped to the image (utils_ms.py:102-105)
__device__ __forceinline__ float4 yl_box_corners(float px, float py, float pw, float ph, float hi) {
#pragma clang fp contract(off)
  float4 bx;
  bx.x = yl_clampf(px - pw * 0.5f, 0.0f, hi);
  bx.y = yl_clampf(py - ph * 0.5f, 0.0f, hi);
  bx.z = yl_clampf(px + pw * 0.5f, 0.0f, hi);
  bx.w = yl_clampf(py + ph * 0.5f, 0.0f, hi);
  return bx;
}
