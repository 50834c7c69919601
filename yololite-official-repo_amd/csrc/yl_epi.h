// Pieces shared by the convolution translation units (yl_conv.hip, yl_convc.hip): pixel descriptor of a lane and
// the float4 epilogues.
#pragma once
#include "yl_internal.h"
#include "yl_dev.h"
#include "yl_decode.h"

// output pixel of a lane
struct YlPix {
  int b, oy, ox;     // output pixel (clamped to a valid pixel for address generation)
  bool valid;        // false for the padding lanes of the last tile (never stored)
  size_t lin;        // linear output pixel index (clamped)
};

// ReLU-family activations as a clamp with wave-uniform bounds; SiLU behind a uniform branch
__device__ __forceinline__ f32x4 yl_actc(f32x4 v, int act, float lo, float hi) {
  if (YL_SMOOTH(act)) return yl_act4(v, act);
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
    float* orow = p.out + px[mt].lin * (p.ldo ? p.ldo : p.N);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = (nt0 + nt) * 16 + 4 * kq;
      f32x4 v = acc[mt][nt];
      if (add_bias) v += yl_ld4(p.bias + n);
      v = yl_clamp4(v, lo, hi);
      if (n < p.N) {
        if (p.ldo) { orow[n] = v.x; orow[n + 1] = v.y; orow[n + 2] = v.z; orow[n + 3] = v.w; }   // rows of ldo floats: unaligned
        else *reinterpret_cast<f32x4*>(orow + n) = v;
      }
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


// Head-output layers under yl_predict: decode fused into the epilogue (replaces yl_decode_score_kernel and the
// 2.9 MB/image write + read of the raw level tensor).  Candidate = pixel of the tile; its row of 5+C logits
// is spread over the wave: channel c = (nt0+nt)*16 + 4*kq + r sits in element r of acc[mt][nt] of lane
// (kq, pl).  tx,ty,tw,th = lane kq 0 / n-tile 0, obj = element 0 of lane kq 1; the class arg-max is a
// per-lane scan + a butterfly over the four kq lanes of the pixel (xor 16, 32).  Class choice follows the
// reference exactly: (conf, idx) = sigmoid(cls).max(-1), FIRST maximum -- i.e. the smallest class whose
// sigmoid equals sigmoid(max logit) (see yl_decode_score_kernel for the band argument).  Same arithmetic
// (yl_decode.h, contraction off) on the same fp32 logits as the unfused path -> bit-identical NMS inputs.
// BIASED: the caller has already added the bias (yl_conv_dpp_kernel keeps the head-output bias in LDS)
template <int NT, int MT, bool BIASED = false>
__device__ __forceinline__ void yl_epi_decode(const YlConvP& p, f32x4 (&acc)[MT][NT], const YlPix (&px)[MT], int nt0,
                                              int kq, int lane) {
#pragma clang fp contract(off)
  const int C = p.dec_C;
  const float* const bias = p.bias;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    f32x4 v[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) v[nt] = BIASED ? acc[mt][nt] : acc[mt][nt] + yl_ld4(bias + (nt0 + nt) * 16 + 4 * kq);
    // ---- objectness: channel 4 = element 0 of the kq-1 lane
    const float tobj = __shfl(v[0].x, (lane & 15) + 16, 64);
    // ---- class logits: local first-maximum, then across the 4 lanes of the pixel
    float lmax = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = (nt0 + nt) * 16 + 4 * kq + r;
        if (ch >= 5 && ch < 5 + C) lmax = fmaxf(lmax, v[nt][r]);
      }
    lmax = fmaxf(lmax, __shfl_xor(lmax, 16, 64));
    lmax = fmaxf(lmax, __shfl_xor(lmax, 32, 64));
    float score;
    int ci = 0;
    const float obj = yl_sigmoid(tobj);
    if (C > 1) {
      const float best = yl_sigmoid(lmax);
      const float band = lmax - 1e-3f * (1.0f + fabsf(lmax));
      const bool wide = best < 1.2e-38f;
      int first = 0x7fffffff;                               // smallest class whose sigmoid equals `best`
      // Fast path (round 5): a logit EQUAL to the maximum has the maximum's sigmoid by construction, so the sigmoid
      // test is only needed for logits strictly below the maximum that pass the band test.  One cheap scan finds the
      // first class equal to the maximum and whether any lane of the wave holds such a near-tie; only then (a
      // wave-uniform branch, rare: another class within 1e-3 relative of the maximum, or saturated logits > 10) the
      // full scan with one sigmoid per candidate runs.  Same predicate, same result; ~24 predicated sigmoids
      // (exp + IEEE division each) per tile leave the common path.
      bool near_tie = false;
#pragma unroll
      for (int nt = NT - 1; nt >= 0; --nt)
#pragma unroll
        for (int r = 3; r >= 0; --r) {
          const int ch = (nt0 + nt) * 16 + 4 * kq + r;
          const float l = v[nt][r];
          if (ch >= 5 && ch < 5 + C) {
            if (l == lmax) first = ch - 5;
            else if (wide || l >= band || l > 10.0f) near_tie = true;
          }
        }
      if (__any(near_tie)) {
        first = 0x7fffffff;
#pragma unroll
        for (int nt = NT - 1; nt >= 0; --nt)
#pragma unroll
          for (int r = 3; r >= 0; --r) {
            const int ch = (nt0 + nt) * 16 + 4 * kq + r;
            const float l = v[nt][r];
            if (ch >= 5 && ch < 5 + C && (wide || l >= band || l > 10.0f) && yl_sigmoid(l) == best) first = ch - 5;
          }
      }
      first = min(first, __shfl_xor(first, 16, 64));
      first = min(first, __shfl_xor(first, 32, 64));
      ci = first;
      score = obj * best;
    } else if (C == 1 && p.dec_mode == YL_POST_FALLBACK) {
      const float l0 = __shfl(v[0].y, (lane & 15) + 16, 64);  // channel 5 = element 1 of the kq-1 lane
      score = obj * yl_sigmoid(l0);
    } else {
      score = obj;
    }
    if (kq == 0 && px[mt].valid) {
      float cx, cy, pw, ph;
      yl_decode_cell((float)px[mt].ox, (float)px[mt].oy, p.dec_stride, v[0].x, v[0].y, v[0].z, v[0].w, p.dec_center,
                     p.dec_wh, cx, cy, pw, ph);
      if (p.dec_mode == YL_POST_FALLBACK && !(pw >= 2.0f && ph >= 2.0f)) score = -INFINITY;
      const size_t o = (size_t)px[mt].b * p.dec_N + p.dec_off + px[mt].oy * p.OW + px[mt].ox;
      p.dec_boxes[o] = yl_box_corners(cx, cy, pw, ph, p.dec_hi);
      p.dec_scores[o] = score;
      p.dec_cls[o] = ci;
    }
  }
}

