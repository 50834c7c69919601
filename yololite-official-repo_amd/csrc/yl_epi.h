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

// store 4 consecutive channels / one channel at ELEMENT offset e of the layer's output tensor.  In the fp16-storage unit the
// tensor is fp16 unless the layer says out_f32 (head outputs -> the fp32 detection levels, the mask prototypes)
__device__ __forceinline__ void yl_out4(const YlConvP& p, size_t e, f32x4 v) {
#if defined(YL_F16S) && YL_F16S
  if (p.out_f32) { yl_st4(reinterpret_cast<float*>(p.out) + e, v); return; }
#endif
  yl_st4(p.out + e, v);
}
__device__ __forceinline__ void yl_out1(const YlConvP& p, size_t e, float v) {
#if defined(YL_F16S) && YL_F16S
  if (p.out_f32) { reinterpret_cast<float*>(p.out)[e] = v; return; }
#endif
  p.out[e] = (yl_act_t)v;
}

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
    const size_t orow = px[mt].lin * (p.ldo ? p.ldo : p.N);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = (nt0 + nt) * 16 + 4 * kq;
      f32x4 v = acc[mt][nt];
      if (add_bias) v += yl_ld4(p.bias + n);
      v = yl_clamp4(v, lo, hi);
      if (n < p.N) {
        if (p.ldo) { yl_out1(p, orow + n, v.x); yl_out1(p, orow + n + 1, v.y); yl_out1(p, orow + n + 2, v.z); yl_out1(p, orow + n + 3, v.w); }   // rows of ldo floats: unaligned
        else yl_out4(p, orow + n, v);
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
      yl_out4(p, obase + n, v);
    }
  }
}


// Head-output layers under yl_predict: decode fused into the epilogue (replaces yl_decode_score_kernel and the
// 2.9 MB/image write + read of the raw level tensor).  Candidate = pixel of the tile; its row of 5+C logits
// is spread over the wave: channel c = (nt0+nt)*16 + 4*kq + r sits in element r of acc[mt][nt] of lane
// (kq, pl).  tx,ty,tw,th = lane kq 0 / n-tile 0, obj = element 0 of lane kq 1.  Class choice follows the
// reference exactly: (conf, idx) = sigmoid(cls).max(-1), FIRST maximum -- i.e. the smallest class whose
// sigmoid equals sigmoid(max logit) (see yl_decode_score_kernel for the band argument).  Same arithmetic
// (yl_decode.h, contraction off) on the same fp32 logits as the unfused path -> bit-identical NMS inputs.
// BIASED: the caller has already added the bias (yl_conv_dpp_kernel keeps the head-output bias in LDS).
//
// Round 6 form.  An ablation of the fused head launch put the decode at 64 of its 246 us -- 4700 SIMD cycles per 16
// candidates, none of them an MFMA, on lanes the fp32 MFMAs share: the first form's class scan compiled into a chain
// of 24 predicated blocks per lane (~25 scalar + vector instructions and two branches each), every lane of a pixel
// evaluated sigmoid(objectness) and sigmoid(max) itself, and the box (two sigmoids, two softplus) ran on the kq-0
// quarter of the lanes.  Now, branch-free on the common path:
//   pass 1   per lane the largest and the second largest of its <= 4 NT class logits in ONE pass (v_max + v_med3 per
//            element; box / objectness / padding channels masked to -inf only in the n-tiles that can hold them: TIGHT);
//            lmax = butterfly over the pixel's four lanes; a lane's near-tie candidate is its largest logit that is not
//            (one instance of) lmax: near tie <=> candidate >= min(band, 10+) -- the predicate of the first form folded
//            into one compare (a second instance of lmax inside a lane also takes the tie path, which is the general one);
//   pass 2   first class equal to lmax: compare + select with a literal per element, butterfly minimum;
//   squash   ONE sigmoid evaluation per wave does tx (lane kq 0), ty (kq 1), objectness (kq 2) and the maximum class
//            logit (kq 3) of 16 candidates at once, ONE softplus / exp evaluation does tw (kq 0) and th (kq 1); the values
//            travel by ds_bpermute.  Every number is produced by the same operation sequence on the same inputs as before
//            (which lane evaluates an expression does not change its bits): bit-identical.
// The tie path (a wave-uniform branch, rare) is the first form's.
__device__ __forceinline__ float yl_vmax(float a, float b) {   // v_max_f32 without the canonicalising self-max that fmaxf
  float d;                                                     // emits in IEEE mode (logits are never NaN here)
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
template <int NT, int MT, bool BIASED = false, bool TIGHT = false>
__device__ __forceinline__ void yl_epi_decode(const YlConvP& p, f32x4 (&acc)[MT][NT], const YlPix (&px)[MT], int nt0,
                                              int kq, int lane) {
#pragma clang fp contract(off)
  const int C = p.dec_C;
  const float* const bias = p.bias;
  const int pl = lane & 15;
  const float NINF = -INFINITY;
  // element e = 16 nt + r of the lane is channel 16 nt0 + 4 kq + e: a class channel iff rlo <= e < rhi
  const int rlo = 5 - 4 * kq - 16 * nt0, rhi = 5 + C - 4 * kq - 16 * nt0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    f32x4 v[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) v[nt] = BIASED ? acc[mt][nt] : acc[mt][nt] + yl_ld4(bias + (nt0 + nt) * 16 + 4 * kq);
    float lmax = 0.0f, cand = NINF;
    int first = 0x7fffffff;
    float cl[NT][4];                                          // class view of the row: everything else is -inf
    if (C > 1) {
      // ---- pass 1: largest / second largest class logit of the lane
      float m1 = NINF, m2 = NINF;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = nt * 16 + r;
          float x = v[nt][r];
          // TIGHT (nt0 == 0, NT <= ceil((5 + C) / 16) + 1: the callers that hold whole rows): only n-tile 0 holds box /
          // objectness channels and only the last two n-tiles can hold padding
          if (!TIGHT || nt == 0) x = e >= rlo ? x : NINF;
          if (!TIGHT || nt >= NT - 2) x = e < rhi ? x : NINF;
          cl[nt][r] = x;
          m2 = __builtin_amdgcn_fmed3f(x, m1, m2);
          m1 = yl_vmax(m1, x);
        }
      lmax = yl_vmax(m1, __shfl_xor(m1, 16, 64));
      lmax = yl_vmax(lmax, __shfl_xor(lmax, 32, 64));
      // ---- pass 2: first class of the lane whose logit EQUALS the maximum (descending scan: the smallest survives)
      int fe = 0x7fffff00;
#pragma unroll
      for (int nt = NT - 1; nt >= 0; --nt)
#pragma unroll
        for (int r = 3; r >= 0; --r) fe = cl[nt][r] == lmax ? nt * 16 + r : fe;
      first = fe - rlo;                                       // channel - 5 (huge when the lane does not hold the maximum)
      cand = m1 == lmax ? m2 : m1;                            // the lane's largest logit that is not (one instance of) the maximum
    }
    // ---- one sigmoid evaluation: tx | ty | objectness | class logit (the maximum; the only class of a C == 1 fallback head)
    const float a_ty = __shfl(v[0].y, pl, 64);
    const float a_obj = __shfl(v[0].x, pl + 16, 64);          // channel 4 = element 0 of the kq-1 lane
    const float a_l0 = __shfl(v[0].y, pl + 16, 64);           // channel 5 = element 1 of the kq-1 lane
    const float s_in = kq == 0 ? v[0].x : (kq == 1 ? a_ty : (kq == 2 ? a_obj : (C > 1 ? lmax : a_l0)));
    const float sg = yl_sigmoid(s_in);
    const float sy = __shfl(sg, pl + 16, 64);
    const float obj = __shfl(sg, pl + 32, 64);
    const float best = __shfl(sg, pl + 48, 64);
    // ---- one width / height evaluation: tw (kq 0) | th (kq 1)
    const float a_th = __shfl(v[0].w, pl, 64);
    const float w_in = kq == 0 ? v[0].z : a_th;
    float wv;
    if (p.dec_wh == YL_WH_SOFTPLUS) {
      wv = yl_softplus(w_in) * p.dec_stride;
    } else if (p.dec_wh == YL_WH_V8) {
      const float a = yl_sigmoid(w_in) * 2.0f;
      wv = (a * a) * p.dec_stride;
    } else {
      wv = expf(yl_clampf(w_in, -4.0f, 4.0f)) * p.dec_stride;
    }
    const float ph = __shfl(wv, pl + 16, 64);
    float score;
    int ci = 0;
    if (C > 1) {
      const float band = lmax - 1e-3f * (1.0f + fabsf(lmax));
      const bool wide = best < 1.2e-38f;
      // a class other than (one instance of) the maximum with  wide || l >= band || l > 10  -- 10+ is the float after 10
      const float thr = wide ? NINF : fminf(band, 0x1.400002p+3f);
      const bool near_tie = cand >= thr;
      if (__any(near_tie)) {                                  // the general scan of the first form: one sigmoid per candidate
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
      score = obj * best;
    } else {
      score = obj;
    }
    if (kq == 0 && px[mt].valid) {
      const float gx = (float)px[mt].ox, gy = (float)px[mt].oy, st = p.dec_stride;
      float cx, cy;
      if (p.dec_center == YL_CENTER_V8) {
        cx = ((sg * 2.0f - 0.5f) + gx) * st;
        cy = ((sy * 2.0f - 0.5f) + gy) * st;
      } else {
        cx = (sg + gx) * st;
        cy = (sy + gy) * st;
      }
      const float pw = wv;
      if (p.dec_mode == YL_POST_FALLBACK && !(pw >= 2.0f && ph >= 2.0f)) score = -INFINITY;
      const size_t o = (size_t)px[mt].b * p.dec_N + p.dec_off + px[mt].oy * p.OW + px[mt].ox;
      p.dec_boxes[o] = yl_box_corners(cx, cy, pw, ph, p.dec_hi);
      p.dec_scores[o] = score;
      p.dec_cls[o] = ci;
    }
  }
}
