// Pieces shared by the convolution translation units (yl_conv.hip, yl_convc.hip): pixel descriptor of a lane and
// the float4 epilogues.
#pragma once
#include "yl_internal.h"
#include "yl_dev.h"

// output pixel of a lane
struct YlPix {
  int b, oy, ox;     // output pixel (clamped to a valid pixel for address generation)
  bool valid;        // false for the padding lanes of the last tile (never stored)
  size_t lin;        // linear output pixel index (clamped)
};

// ReLU-family activations as a clamp with wave-uniform bounds; SiLU behind a uniform branch
__device__ __forceinline__ f32x4 yl_actc(f32x4 v, int act, float lo, float hi) {
  if (act == YL_ACT_SILU) return yl_act4(v, YL_ACT_SILU);
  return yl_clamp4(v, lo, hi);
}

__device__ __forceinline__ f32x4 yl_sel4(bool keep, f32x4 v) {
  f32x4 r;
  r.x = keep ? v.x : 0.f; r.y = keep ? v.y : 0.f; r.z = keep ? v.z : 0.f; r.w = keep ? v.w : 0.f;
  return r;
}

// ---- epilogues.  Lane holds channels n..n+3 (n = ntile*16 + 4*kq) of pixel px[mt].
// ReLU-family activations are a branch-free clamp to [lo,hi] (lo=-inf/0, hi=6/+inf).
// add_bias = false: the accumulators were initialised with the bias (no residual pre-add in the way)
template <int NT, int MT>
__device__ __forceinline__ void yl_epi_fast(const YlConvP& p, f32x4 (&acc)[MT][NT], const YlPix (&px)[MT], int nt0,
                                            int kq, float lo, float hi, bool add_bias) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (!px[mt].valid) continue;
    float* orow = p.out + px[mt].lin * p.N;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = (nt0 + nt) * 16 + 4 * kq;
      f32x4 v = acc[mt][nt];
      if (add_bias) v += yl_ld4(p.bias + n);
      v = yl_clamp4(v, lo, hi);
      if (n < p.N) *reinterpret_cast<f32x4*>(orow + n) = v;
    }
  }
}

template <int NT, int MT>
__device__ __forceinline__ void yl_epi_generic(const YlConvP& p, f32x4 (&acc)[MT][NT], const YlPix (&px)[MT],
                                               int nt0, int kq) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (!px[mt].valid) continue;
    const size_t obase = px[mt].lin * p.N;
    size_t up_off = 0;
    if (p.up) {
      const int uy = (px[mt].oy * p.UH) / p.OH, ux = (px[mt].ox * p.UW) / p.OW;
      up_off = (((size_t)px[mt].b * p.UH + uy) * p.UW + ux) * p.N;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = (nt0 + nt) * 16 + 4 * kq;
      if (n >= p.N) continue;
      f32x4 v = acc[mt][nt] + yl_ld4(p.bias + n);
      v = yl_act4(v, p.act);
      if (p.res) v += yl_ld4(p.res + obase + n);
      if (p.up) v += yl_ld4(p.up + up_off + n);
      *reinterpret_cast<f32x4*>(p.out + obase + n) = v;
    }
  }
}

