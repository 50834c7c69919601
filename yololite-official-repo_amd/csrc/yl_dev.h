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
__device__ __forceinline__ void yl_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// fp16 activation tensors (the fp16-storage unit, yl_internal.h: yl_act_t): four consecutive channels are ONE 8-byte access;
// arithmetic stays fp32 (loads widen exactly, stores round to nearest even like torch's autocast casts)
typedef _Float16 yl_h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 yl_ld4(const _Float16* p) {
  return __builtin_convertvector(*reinterpret_cast<const yl_h16x4*>(p), f32x4);
}
__device__ __forceinline__ void yl_st4(_Float16* p, f32x4 v) {
  *reinterpret_cast<yl_h16x4*>(p) = __builtin_convertvector(v, yl_h16x4);
}


// clamp to [lo,hi] in ONE VALU op per element (v_med3_f32).  fp32 MFMA and fp32 VALU share the SIMD's FMA
// lanes on gfx950 (same 64 FLOP/clk/SIMD peak; measured: removing VALU work shortens MFMA-bound kernels 1:1),
// so epilogue instruction count is kernel time.  lo = -inf / hi = +inf give the one-sided / identity cases.
// acc + a * b per component as two v_pk_fma_f32 (IEEE fma per lane and component, the bits of four fmaf calls, half the
// VALU issue time).  On gfx950 the fp32 MFMA runs on the vector FMA lanes (its peak is the vector peak), and kernels whose
// VALU count was cut got faster by about the cut even when MFMA-bound -- so VALU instructions saved in a depthwise inner
// loop are MFMA time gained.
__device__ __forceinline__ f32x4 yl_fma4(f32x4 a, f32x4 b, f32x4 acc) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ lo = __builtin_elementwise_fma((f32x2_){a.x, a.y}, (f32x2_){b.x, b.y}, (f32x2_){acc.x, acc.y});
  const f32x2_ hi = __builtin_elementwise_fma((f32x2_){a.z, a.w}, (f32x2_){b.z, b.w}, (f32x2_){acc.z, acc.w});
  return (f32x4){lo.x, lo.y, hi.x, hi.y};
}

// B fragments of two Winograd positions from the six window reads of an m-tile (yl_conv_wino2_kernel): u_c = y_c * sr + x_c,
// b0 = u0 - u1, b1 = u2 * sc + u1 as TEN packed instructions by name.  Inline asm because the instruction selector splits the
// builtin form into scalar v_fma_f32 / v_sub_f32 pairs in register-tight loops (61 instead of 40 VALU instructions per k-block),
// and on gfx950 a VALU instruction of one wave is matrix-pipe time of the SIMD's other wave.  The trailing s_nop 1: b0 / b1 are
// MFMA operands, a VALU result needs two wait states before a matrix instruction reads it, and the hazard recognizer does not
// look into asm statements (without it: wrong results).  fma(u1, -1, u0) would be the same bits as u0 - u1; v_pk_add_f32 with
// the negate modifier needs no constant register.
__device__ __forceinline__ void yl_wino_b(const f32x4 (&x)[3], const f32x4 (&y)[3], f32x4 sr4, f32x4 sc4, f32x4& b0, f32x4& b1) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  f32x2_ u[3][2];
  const f32x2_ sr = (f32x2_){sr4.x, sr4.y}, sc = (f32x2_){sc4.x, sc4.y};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(u[c][0]) : "v"((f32x2_){y[c].x, y[c].y}), "v"(sr), "v"((f32x2_){x[c].x, x[c].y}));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(u[c][1]) : "v"((f32x2_){y[c].z, y[c].w}), "v"(sr), "v"((f32x2_){x[c].z, x[c].w}));
  }
  f32x2_ r0, r1, r2, r3;
  asm("v_pk_add_f32 %0, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %1, %5, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %2, %8, %10, %6\n\t"
      "v_pk_fma_f32 %3, %9, %10, %7\n\t"
      "s_nop 1"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"(u[0][0]), "v"(u[0][1]), "v"(u[1][0]), "v"(u[1][1]), "v"(u[2][0]), "v"(u[2][1]), "v"(sc));
  b0 = (f32x4){r0.x, r0.y, r1.x, r1.y};
  b1 = (f32x4){r2.x, r2.y, r3.x, r3.y};
}

// acc + a * b per component as two v_pk_fma_f32 BY NAME (the builtin form above is split into scalar instructions when the
// selector likes).  Inline asm: the hazard recognizer does not see a VALU write in it -- the result must not be an operand of a
// matrix instruction within two wait states (yl_conv_dwl_kernel: it is the next k-block's B fragment).
__device__ __forceinline__ f32x4 yl_pk_fma4(f32x4 a, f32x4 b, f32x4 acc) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  f32x2_ lo, hi;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"((f32x2_){a.x, a.y}), "v"((f32x2_){b.x, b.y}), "v"((f32x2_){acc.x, acc.y}));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"((f32x2_){a.z, a.w}), "v"((f32x2_){b.z, b.w}), "v"((f32x2_){acc.z, acc.w}));
  return (f32x4){lo.x, lo.y, hi.x, hi.y};
}

__device__ __forceinline__ f32x4 yl_clamp4(f32x4 v, float lo, float hi) {
  f32x4 r;
  r.x = __builtin_amdgcn_fmed3f(v.x, lo, hi); r.y = __builtin_amdgcn_fmed3f(v.y, lo, hi);
  r.z = __builtin_amdgcn_fmed3f(v.z, lo, hi); r.w = __builtin_amdgcn_fmed3f(v.w, lo, hi);
  return r;
}

// ---- MFMA step: one 16-channel k-block, NT n-tiles x MT m-tiles.  Lane (kq = lane >> 4, i = lane & 15) holds
// four consecutive channels 4kq..4kq+3 of its row in both operands (the transposed-GEMM layout of yl_conv.hip).
//   fp32 build:  4 x v_mfma_f32_16x16x4_f32 per (nt, mt)  -- exact fp32 (the parity path)
//   YL_BF16 build (second compilation of the conv translation units, SURVEY 8(f) f4 "bf16 MFMA inference
//   mode"): operands rounded to bf16 in registers (v_cvt_pk_bf16_f32, RNE), ONE v_mfma_f32_16x16x16_bf16 per
//   (nt, mt), fp32 accumulate; activations and weights stay fp32 in HBM / LDS, so layouts, loads and epilogues
//   are shared with the fp32 build.  Selected per context with yl_set_option("mfma_bf16", 1).
//   YL_F16 (third compilation, -DYL_BF16=1 -DYL_F16=1: everything the reduced-precision build shares, plus): operands rounded to
//   fp16 (RNE) and multiplied on v_mfma_f32_16x16x16_f16 -- the counterpart of the reference's fp16 autocast in evaluate_model
//   (scripts/helpers/evaluate.py:399,415).  Selected with yl_set_option("mfma_f16", 1).  Symbols carry _f16 (yl_lp.h).
#ifndef YL_BF16
#define YL_BF16 0
#endif
#ifndef YL_F16
#define YL_F16 0
#endif
#include "yl_lp.h"
typedef short yl_s16x4 __attribute__((ext_vector_type(4)));
// the lane's four consecutive channels packed to the 16-bit operand type of the reduced-precision build (bf16 or fp16, RNE)
__device__ __forceinline__ yl_s16x4 yl_pk_bf16(f32x4 v) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  u32x2_ r;
#if YL_F16
  typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
  r.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v.x, v.y}, f16x2_));
  r.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v.z, v.w}, f16x2_));
#else
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  r.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v.x, v.y}, bf16x2_));
  r.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){v.z, v.w}, bf16x2_));
#endif
  return __builtin_bit_cast(yl_s16x4, r);
}
// one 16x16x16 MFMA on the packed operands, fp32 accumulate
#if YL_F16
typedef _Float16 yl_f16x4 __attribute__((ext_vector_type(4)));
#define YL_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(yl_f16x4, a), __builtin_bit_cast(yl_f16x4, b), c, 0, 0, 0)
#else
#define YL_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)
#endif
#if YL_BF16
#define YL_MFMA_PER_BLOCK 1
#define yl_mma_step YL_LP_NAME(yl_mma_step)
#else
#define YL_MFMA_PER_BLOCK 4
#endif
template <int NT, int MT>
__device__ __forceinline__ void yl_mma_step(const f32x4 (&wq)[NT], const f32x4 (&xq)[MT], f32x4 (&acc)[MT][NT]) {
#if YL_BF16
  yl_s16x4 xb[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) xb[mt] = yl_pk_bf16(xq[mt]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const yl_s16x4 wb = yl_pk_bf16(wq[nt]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      acc[mt][nt] = YL_MFMA16(wb, xb[mt], acc[mt][nt]);
  }
#else
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[nt][s], xq[mt][s], acc[mt][nt], 0, 0, 0);
#endif
}

// ---- asynchronous global -> LDS copies (global_load_lds_dword[x4]): no VGPR staging and, unlike a load followed
// by a ds_write, nothing in the issuing wave waits for the data -- the weight images of a workgroup land in LDS
// while the wave goes on to issue its first activation loads.  (With load + ds_write the weight fetch sat in front
// of every launch's first tile: measured 0.18 ms of a 2.08 ms step over the 40 launches.)  `lds_row` is wave-uniform;
// lane l's 16 (4) bytes land at lds_row + 16*l (4*l).  Completion: s_waitcnt vmcnt(0), which the compiler places in
// front of the next workgroup barrier.
__device__ __forceinline__ void yl_glds16(const void* g_lane, void* lds_row) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane,
                                   (__attribute__((address_space(3))) void*)lds_row, 16, 0, 0);
}
__device__ __forceinline__ void yl_glds4(const void* g_lane, void* lds_row) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane,
                                   (__attribute__((address_space(3))) void*)lds_row, 4, 0, 0);
}
// `n` floats (any n >= 0) from global to LDS with the whole workgroup; tail lanes are masked
__device__ __forceinline__ void yl_glds_floats(const float* g, float* lds, int n, int tid, int nthreads) {
  const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
  for (int i0 = wave * 64; i0 < n; i0 += nw * 64)
    if (i0 + lane < n) yl_glds4(g + i0 + lane, lds + i0);
}
