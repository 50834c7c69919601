// Fused network entry block for gfx950:  stem 3x3 s2 (3 -> C1)  ->  3x3 s2 pad 1 (C1 -> C2)  ->  optional 1x1
// (C2 -> C3), NCHW fp32 input to NHWC fp32 output.
//
// Why: the stem's output (S/2 x S/2 x C1, 13.1 MB per 640x640 image for C1 = 32) is the largest tensor of
// the network; written and re-read it costs 26 MB/image of HBM traffic, more than the rest of edge_n's
// layer-fused traffic together.  Here it lives only in LDS: one WAVE produces a 2x8 tile of the second
// conv's output from a 5x17 patch of stem pixels kept in its private LDS region.
//
// Phase 1  stem on the 5x16 new pixels of the patch: transposed MFMA GEMM (A = stem weights in registers, 7 k-steps,
//          B = one input scalar per lane gathered from the NCHW planes), result (+bias, act; zero outside
//          the image = the second conv's zero padding) -> LDS patch [5][17][C1+4].
// Phase 2  3x3 s2 conv from LDS: one 16-pixel m-tile per wave, B operand = float4 of 4 channels read
//          from the patch, A operand = packed weights in LDS (one ds_read_b128 per 16x16x16 step).
// Phase 3  optional 1x1 conv chained in registers: the MFMA D layout (lane = pixel, 4 consecutive
//          channels) IS the B-operand layout of the next GEMM's k-block, so no data movement at all.
//
// Replaces timm conv_stem+bn1 and blocks.0.{0,1} (ConvBnAct) of mobilenetv4_conv_small*, i.e. the
// first three conv/BN/ReLU triples behind model_v2.py:94-100,266-272.
// bf16-MFMA variant: second compilation with -DYL_BF16=1 under distinct symbol names (see yl_dev.h: yl_mma_step)
#include "yl_lp.h"
#if defined(YL_BF16) && YL_BF16
#define yl_stemblock_kernel YL_LP_NAME(yl_stemblock_kernel)
#define yl_launch_stemblock YL_LP_NAME(yl_launch_stemblock)
#define yl_stemblock_init YL_LP_NAME(yl_stemblock_init)
#define yl_stemblock_supported YL_LP_NAME(yl_stemblock_supported)
#define yl_stemdw_kernel YL_LP_NAME(yl_stemdw_kernel)
#define yl_launch_stemdw YL_LP_NAME(yl_launch_stemdw)
#endif
#include "yl_internal.h"
#include "yl_dev.h"
#include <type_traits>

#define SB_TR 2                     // wave tile: 2 rows x 8 columns of the second conv's output grid
#define SB_TC 8
#define SB_PR (2 * SB_TR + 1)       // stem patch: 5 rows x 17 columns
#define SB_PC (2 * SB_TC + 1)
#define SB_NPATCH (SB_PR * SB_PC)   // 85 stem pixels
#define SB_NM SB_PR                 // stem m-tiles per tile: patch row m x columns 1..16 (column 0 rolls over, see below)

// Every WAVE owns its tiles end to end (private LDS patch, no workgroup barrier in the loop): waves
// drift apart and cover each other's gather latency / MFMA dependency stalls.
//
// Round 3: a wave walks a horizontal STRIP of tiles left to right.  The 5 x 17 stem patch of a tile shares its
// column 0 with column 16 of the tile before: that column stays in LDS (copied 16 -> 0, five pixels), and the stem
// GEMM covers the 5 x 16 NEW pixels = exactly five 16-pixel m-tiles with no padding rows -- 70 stem MFMAs per tile
// where the stand-alone 85-pixel patch (padded to 96) took 84, 146 instead of 160 per tile in total.  Only the first
// tile of a strip computes its own column 0 (one more m-tile with five live rows).  Further VALU work removed from
// the MFMA-issue-bound loop (fp32 MFMA and fp32 VALU share the FMA lanes, see yl_dev.h): the stem bias rides in the
// K = 27 -> 28 pad slot of the stem GEMM (weight slot = bias, input slot = 1.0: added last in the fma chain, the same
// rounding as the separate add), and all tile bookkeeping is scalar (wave id through readfirstlane), so the gathers are
// `global_load saddr + per-lane constant offset` with no per-tile vector address arithmetic.
#define SB_SROWS (2 * (SB_PR - 1) + 3)     // input rows under a patch (stem stride 2, 3x3): 11
#define SB_SCOLS (2 * (SB_PC - 1) + 3)     // input columns: 35
#define SB_NSEG (3 * SB_SROWS)             // (channel, row) segments: 33
#define SB_SP 36                           // column pitch of the staged block: 9 chunks of 4 floats per segment
#define SB_NCH (SB_NSEG * SB_SP / 4)       // 16-byte chunks: 297
#define SB_NSTG ((SB_NCH + 63) / 64)       // LDS-DMA instructions (64 lanes x 16 bytes) per tile: 5
#define SB_NSTG1 ((SB_NSEG * SB_SP + 63) / 64)   // dword-granular rows (image-border tiles): 19
template <int NT1 /*C1/16*/, int NT2 /*ceil(C2/16)*/, int NT3 /*ceil(C3/16), 0 = no 1x1*/, int NWV /*waves per workgroup*/>
__global__ __launch_bounds__(NWV * 64, 2) void yl_stemblock_kernel(YlConvP p) {
  constexpr int C1 = NT1 * 16, P1 = C1 + 4, KS = 7, KB1 = NT1;
  constexpr int PATCH_F = SB_NPATCH * P1;                            // floats per wave patch
  constexpr int WAVE_F = PATCH_F + SB_NSTG * 256;                     // + the staged input of the tile's new pixels
  extern __shared__ __attribute__((aligned(16))) float sb_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  float* patch = sb_lds + wave * WAVE_F;                             // [5][17][P1], wave private
  float* stage = patch + PATCH_F;                                    // [33 segments][36 columns] (+ tail of the 5th KB)
  f32x4* w2l = reinterpret_cast<f32x4*>(sb_lds + NWV * WAVE_F);
  f32x4* w3l = w2l + 9 * KB1 * NT2 * 64;

  // ---- once per block: stem A fragments -> registers, conv2 / conv3 weights -> LDS
  float wa[KS][NT1];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) wa[s][nt] = p.wp[(s * NT1 + nt) * 64 + lane];
#if YL_BF16
  // bf16 operands would round the bias: this build keeps the separate fp32 add and a zero weight in the pad slot
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) if (kq == 3) wa[6][nt] = 0.0f;
  yl_s16x4 wab[2][NT1];
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) {
    wab[0][nt] = yl_pk_bf16((f32x4){wa[0][nt], wa[1][nt], wa[2][nt], wa[3][nt]});
    wab[1][nt] = yl_pk_bf16((f32x4){wa[4][nt], wa[5][nt], wa[6][nt], 0.0f});
  }
  f32x4 bias1[NT1];
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) bias1[nt] = yl_ld4(p.bias + nt * 16 + 4 * kq);
#endif
  {
    const f32x4* g2 = reinterpret_cast<const f32x4*>(p.w2p);
    for (int r = wave; r < 9 * KB1 * NT2; r += NWV) yl_glds16(g2 + r * 64 + lane, w2l + r * 64);   // async, see yl_dev.h
    if (NT3 > 0) {
      const f32x4* g3 = reinterpret_cast<const f32x4*>(p.w3p);
      for (int r = wave; r < NT2 * NT3; r += NWV) yl_glds16(g3 + r * 64 + lane, w3l + r * 64);
    }
  }
  __syncthreads();

  const int plane = p.H * p.W;
  // per-lane constants: tap decode of the lane's k slots.  K order (shared with pack_stem_rows in yl_api.hip): the 27
  // taps are 9 rows (c,ky) of 3 consecutive kx.  Lane group kq owns rows 2kq and 2kq+1 whole (slots 0-2, 3-5) and one
  // element of row 8 (slot 6, kx = kq; group 3: the bias slot), so a lane's 7 operands per patch pixel are two 12-byte
  // loads and one 4-byte load instead of seven scattered dwords.
  int tky[KS], tkx[KS], tc[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    int row, kx;
    if (s < 3) { row = 2 * kq; kx = s; }
    else if (s < 6) { row = 2 * kq + 1; kx = s - 3; }
    else if (kq < 3) { row = 8; kx = kq; }
    else { row = 7; kx = 2; }                                        // bias slot: any valid address, value replaced by 1.0
    tc[s] = row / 3;
    tky[s] = row - 3 * tc[s];
    tkx[s] = kx;
  }
  // Round 3 (second half): the input of a tile's new stem pixels is STAGED in the wave's LDS region by asynchronous
  // global -> LDS copies -- five 16-byte-per-lane loads cover the [33 (channel,row) segments][36 columns] block (lane =
  // one 4-float chunk of one segment; no VGPRs, no ds_write) -- instead of ten 12-byte + seven 4-byte scattered gathers
  // into registers: the ablations put 19 % of the kernel on the vector-memory path of those gathers.  Phase 1 reads its
  // B operands from the stage (7 ds_read_b32 per m-tile).  Column 35 of a segment is not used; interior tiles may read
  // it one float past a row's end (never past the tensor: see `interior` in gather()).
  int goff[SB_NSTG];                                                 // global offset (floats) of chunk k*64 + lane
#pragma unroll
  for (int k = 0; k < SB_NSTG; ++k) {
    int e = k * 64 + lane;
    e = e < SB_NCH ? e : SB_NCH - 1;
    const int seg = e / (SB_SP / 4), ch = e - seg * (SB_SP / 4);
    const int c = seg / SB_SROWS, r = seg - c * SB_SROWS;
    goff[k] = c * plane + r * p.W + 4 * ch;
  }
  const int prow = (pl < SB_PR ? pl : SB_PR - 1);
  // stage index of the lane's k slot s: regular m-tile m (patch row m, column 1 + pl) = rs[s] + 72 m; first-tile m-tile
  // (patch column 0, row prow) = rf[s]
  int rs[KS], rf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    rs[s] = (tc[s] * SB_SROWS + tky[s]) * SB_SP + tkx[s] + 2 * (1 + pl);
    rf[s] = (tc[s] * SB_SROWS + 2 * prow + tky[s]) * SB_SP + tkx[s];
  }
#if YL_BF16
  const bool slot6_lane = true;                                      // bf16 build: separate bias add, zero weight in the slot
#else
  const bool slot6_lane = kq != 3;                                   // lane group 3: the bias slot (input 1.0)
#endif
  f32x4 bias2[NT2];
#pragma unroll
  for (int nt = 0; nt < NT2; ++nt) bias2[nt] = yl_ld4(p.b2 + nt * 16 + 4 * kq);
  // ReLU-family activations as branch-free clamps (SiLU is rejected for this op at yl_create)
  const float lo1 = (p.act == YL_ACT_NONE) ? -INFINITY : 0.0f, hi1 = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float lo2 = (p.act2 == YL_ACT_NONE) ? -INFINITY : 0.0f, hi2 = (p.act2 == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float lo3 = (p.act3 == YL_ACT_NONE) ? -INFINITY : 0.0f, hi3 = (p.act3 == YL_ACT_RELU6) ? 6.0f : INFINITY;
  auto clamp4 = [](f32x4 v, float lo, float hi) { return yl_clamp4(v, lo, hi); };
  f32x4 bias3[NT3 > 0 ? NT3 : 1];
#pragma unroll
  for (int nt = 0; nt < NT3; ++nt) bias3[nt] = yl_ld4(p.b3 + nt * 16 + 4 * kq);
  const int lqr = (1 + pl) * P1 + 4 * kq;                            // patch write offset: regular m-tile (+ m * 17 * P1)
  const int lqf = prow * SB_PC * P1 + 4 * kq;                        // patch write offset: column-0 m-tile
  const int ty = pl >> 3, tx = pl & 7;                               // lane's pixel inside the 2x8 tile
  const int lr = ((2 * ty) * SB_PC + 2 * tx) * P1 + 4 * kq;          // lane part of the patch read offset
  const int Nout = (NT3 > 0) ? p.C3 : p.C2;
  const int tpr = (p.OW + SB_TC - 1) / SB_TC, tpc = (p.OH + SB_TR - 1) / SB_TR;
  const int SL = p.sb_strip;                                         // tiles per strip
  const int spr = (tpr + SL - 1) / SL;                               // strips per tile row
  const int strips_img = spr * tpc;
  const int nstrips = p.B * strips_img;

  // XCD-aware order: workgroup b runs on XCD b % 8 (observed dispatch order; placement changes speed only), and XCD x
  // owns the contiguous strip range [x*nstrips/8, (x+1)*nstrips/8) -- whole images for B % 8 == 0 -- so the input rows
  // shared by vertically neighbouring patches are re-read from this XCD's L2 instead of from HBM a second time.
  int strip = blockIdx.x * NWV + wave, sstride = gridDim.x * NWV, send = nstrips;
  if ((gridDim.x & 7) == 0) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, nj = gridDim.x >> 3;
    const int r0 = (int)(((long)nstrips * x) >> 3), r1 = (int)(((long)nstrips * (x + 1)) >> 3);
    strip = r0 + j * NWV + wave; sstride = nj * NWV; send = r1;
  }
  // the tile whose inputs are being gathered (one ahead of the tile being computed); everything here is wave-uniform
  int g_b = 0, g_ty = 0, g_tx = 0, g_txend = 0, g_first = 0, g_valid = 0, g_strip = strip;
  auto decode_strip = [&]() {
    g_valid = g_strip < send;
    if (!g_valid) return;
    g_b = g_strip / strips_img;
    const int rem = g_strip - g_b * strips_img;
    g_ty = rem / spr;
    g_tx = (rem - g_ty * spr) * SL;
    g_txend = (g_tx + SL) < tpr ? (g_tx + SL) : tpr;
    g_first = 1;
  };
  auto advance = [&]() {
    if (g_tx + 1 < g_txend) { ++g_tx; g_first = 0; }
    else { g_strip += sstride; decode_strip(); }
  };

  // staging of the input block under the patch of tile (g_b, g_ty, g_tx): issued for tile t+1 during tile t's 3x3 phase
  // (one or two LDS-DMA rows per MFMA step), consumed by tile t+1's stem phase after an s_waitcnt vmcnt(0).
  const float* const xnet = reinterpret_cast<const float*>(p.x);   // the network input: fp32 NCHW in every build
  const float* const zf = reinterpret_cast<const float*>(p.zeros);
  const float* g_xo = xnet;                           // interior staging in flight: uniform base of its input block
  bool g_int = false;
  auto gather_piece = [&](int k) { yl_glds16(g_xo + goff[k], stage + k * 256); };
  auto gather = [&]() {
    const int sy0 = 2 * g_ty * SB_TR - 1, sx0 = 2 * g_tx * SB_TC - 1;
    const int iy0 = sy0 * p.stride - p.pad_t, ix0 = sx0 * p.stride - p.pad_l;
    const float* xb = xnet + (size_t)g_b * 3 * plane;
    const bool interior = sy0 >= 0 && sx0 >= 0 && sy0 + SB_PR <= p.SH && sx0 + SB_PC <= p.SW && iy0 >= 0 && ix0 >= 0 &&
                          iy0 + SB_SROWS <= p.H && ix0 + SB_SCOLS <= p.W &&
                          (ix0 + SB_SP <= p.W || iy0 + SB_SROWS < p.H || g_b + 1 < p.B);   // column 35: inside the tensor
    g_int = interior;
    if (interior) {                                   // the common case: uniform base + per-lane constant offsets; the
      g_xo = xb + (long)iy0 * p.W + ix0;              // loads themselves go out inside phase 2 (gather_piece)
    } else {                                          // image border: per-lane bounds, out-of-image elements read zeros
#pragma unroll
      for (int k = 0; k < SB_NSTG1; ++k) {
        int e = k * 64 + lane;
        e = e < SB_NSEG * SB_SP ? e : SB_NSEG * SB_SP - 1;
        const int seg = e / SB_SP, col = e - seg * SB_SP;
        const int c = seg / SB_SROWS, r = seg - c * SB_SROWS;
        const int iy = iy0 + r, ix = ix0 + col;
        const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        yl_glds4(in ? xb + c * plane + (long)iy * p.W + ix : zf, stage + k * 64);
      }
    }
  };
#ifdef SB_STAGGER
  if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(SB_STAGGER);   // de-phase the two co-resident blocks of a CU
#endif
  decode_strip();
  if (g_valid) {
    gather();
    if (g_int) {
#pragma unroll
      for (int k = 0; k < SB_NSTG; ++k) gather_piece(k);
    }
  }
  // the output stores of tile t are issued at the start of tile t+1's 3x3 phase, IN FRONT of that phase's staging loads:
  // vmcnt counts loads and stores in one in-order counter, and the vmcnt(0) in front of a stem phase must not sit
  // behind stores issued a moment earlier
  constexpr int NTO = NT3 > 0 ? NT3 : NT2;
  f32x4 ov[NTO];
  yl_act_t* o_row = p.out;
  bool o_valid = false;
  auto flush_out = [&]() {
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt) {
      const int n = nt * 16 + 4 * kq;
      if (o_valid && n < Nout) yl_st4(o_row + n, ov[nt]);
    }
  };

  while (g_valid) {
    const int b = g_b, tyi = g_ty, txi = g_tx, first = g_first;       // the tile computed now (its inputs are staged)
    const int oy0 = tyi * SB_TR, ox0 = txi * SB_TC;                  // tile origin on the conv2 output grid
    const int sy0 = 2 * oy0 - 1, sx0 = 2 * ox0 - 1;                  // patch origin on the stem grid
    const bool interior = sy0 >= 0 && sx0 >= 0 && sy0 + SB_PR <= p.SH && sx0 + SB_PC <= p.SW;
    // ---- phase 0: column 16 of the previous tile's patch is column 0 of this one (five pixels x C1 channels)
    if (!first) {
      constexpr int Q = C1 / 4;                                      // float4 per pixel
      if (lane < SB_PR * Q) {
        const int i = lane / Q, c4 = lane - i * Q;
        const f32x4 v = *reinterpret_cast<const f32x4*>(patch + (i * SB_PC + (SB_PC - 1)) * P1 + 4 * c4);
        *reinterpret_cast<f32x4*>(patch + (i * SB_PC) * P1 + 4 * c4) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // ---- phase 1: stem on the new patch pixels -> wave-private LDS.  Two copies of the code, selected by a wave-uniform
    // branch: interior tiles (all but the image border) carry no padding logic at all -- left as a runtime flag
    // the compiler predicates it per lane (compares, exec masking and 8 v_cndmask per m-tile on every tile)
    auto stem_mtile = [&](auto interior_tag, const int (&ri)[KS], int roff, int sy, int sx, float* dst, bool live) {
      constexpr bool INTERIOR = decltype(interior_tag)::value;
      float xs[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) xs[s] = stage[ri[s] + roff];
      f32x4 a1[NT1];
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) a1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if YL_BF16
      {   // the lane's 7 k slots as two 4-wide bf16 operands (slot 7 = zero); same slot <-> lane pairing in A and B
        const yl_s16x4 x0 = yl_pk_bf16((f32x4){xs[0], xs[1], xs[2], xs[3]});
        const yl_s16x4 x1 = yl_pk_bf16((f32x4){xs[4], xs[5], xs[6], 0.0f});
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) {
          a1[nt] = YL_MFMA16(wab[0][nt], x0, a1[nt]);
          a1[nt] = YL_MFMA16(wab[1][nt], x1, a1[nt]);
        }
      }
#else
      // the bias slot of the stem GEMM: weight = bias (pack_stem_rows), input = 1.0 on lane group 3
      const float x6 = slot6_lane ? xs[6] : 1.0f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
          a1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][nt], s == 6 ? x6 : xs[s], a1[nt], 0, 0, 0);
#endif
      bool inside = true;
      if (!INTERIOR) inside = sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW;   // else: zero padding of the second conv
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) {
#if YL_BF16
        f32x4 v = clamp4(a1[nt] + bias1[nt], lo1, hi1);     // conv + shift, the reference's order
#else
        f32x4 v = clamp4(a1[nt], lo1, hi1);                 // the shift was the last product of the fma chain
#endif
        if (!INTERIOR && !inside) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (live) *reinterpret_cast<f32x4*>(dst + nt * 16) = v;
      }
    };
    auto phase1 = [&](auto interior_tag) {
      if (first) stem_mtile(interior_tag, rf, 0, sy0 + prow, sx0, patch + lqf, pl < SB_PR);
#pragma unroll
      for (int m = 0; m < SB_NM; ++m)
        stem_mtile(interior_tag, rs, m * 2 * SB_SP, sy0 + m, sx0 + 1 + pl, patch + m * SB_PC * P1 + lqr, true);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the staged input has landed (LDS-DMA completion)
    if (interior) phase1(std::true_type{});
    else phase1(std::false_type{});
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // LDS writes of other lanes -> reads below
    __builtin_amdgcn_wave_barrier();
    advance();
    if (g_valid) gather();                                           // next tile's inputs: in flight during phases 2-3

    // ---- phase 2: 3x3 stride-2 conv on the wave's 2x8 tile.  LDS operands of step i+1 are requested
    //      before the MFMAs of step i (explicit double buffer, order pinned with sched_group_barrier), and
    //      each n-tile accumulates in two chains (the 16x16x4 f32 MFMA has a 40-cycle dependent latency
    //      against a 32-cycle issue interval).
    f32x4 a2[NT2], a2b[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) { a2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; a2b[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    constexpr int NSTEP = 9 * KB1;
    constexpr int PPS = (SB_NSTG + NSTEP - 1) / NSTEP;               // staging rows per MFMA step
    f32x4 xq[2], wq[2][NT2];
    auto lds_step = [&](int i, int buf) {              // i = tap * KB1 + kb  (all compile-time after unrolling)
      const int tap = i / KB1, kb = i - tap * KB1;
      const int ky = tap / 3, kx = tap - 3 * ky;
      xq[buf] = *reinterpret_cast<const f32x4*>(patch + lr + (ky * SB_PC + kx) * P1 + kb * 16);
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) wq[buf][nt] = w2l[(i * NT2 + nt) * 64 + lane];
    };
    auto phase2 = [&](auto gather_tag) {
      constexpr bool GATHER = decltype(gather_tag)::value;           // the next tile's 12-byte gathers ride along
      lds_step(0, 0);
#pragma unroll
      for (int i = 0; i < NSTEP; ++i) {
        if (i + 1 < NSTEP) lds_step(i + 1, (i + 1) & 1);
        int npiece = 0;
        if (GATHER) {
#pragma unroll
          for (int k = 0; k < PPS; ++k)
            if (i * PPS + k < SB_NSTG) { gather_piece(i * PPS + k); ++npiece; }
        }
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
          const f32x4 w = wq[i & 1][nt], x = xq[i & 1];
#if YL_BF16
          if (i & 1) a2b[nt] = YL_MFMA16(yl_pk_bf16(w), yl_pk_bf16(x), a2b[nt]);
          else a2[nt] = YL_MFMA16(yl_pk_bf16(w), yl_pk_bf16(x), a2[nt]);
#else
          a2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], x[0], a2[nt], 0, 0, 0);
          a2b[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], x[1], a2b[nt], 0, 0, 0);
          a2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], x[2], a2[nt], 0, 0, 0);
          a2b[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], x[3], a2b[nt], 0, 0, 0);
#endif
        }
#if !YL_BF16
        __builtin_amdgcn_sched_group_barrier(0x100, 1 + NT2, 0);     // DS reads of step i+1
        if (npiece) __builtin_amdgcn_sched_group_barrier(0x020, PPS, 0);   // this step's gather pieces (VMEM reads)
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT2, 0);     // MFMAs of step i
#endif
      }
    };
    flush_out();                                                     // the previous tile's outputs
    if (g_valid && g_int) phase2(std::true_type{});
    else phase2(std::false_type{});
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) a2[nt] = clamp4((a2[nt] + a2b[nt]) + bias2[nt], lo2, hi2);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // patch reads done before the next tile's writes
    __builtin_amdgcn_wave_barrier();

    // ---- phase 3: optional 1x1 conv chained in registers; the store is deferred (flush_out)
    const int oy = oy0 + ty, ox = ox0 + tx;
    o_valid = oy < p.OH && ox < p.OW;
    o_row = p.out + (((size_t)b * p.OH + oy) * p.OW + ox) * Nout;
    if constexpr (NT3 > 0) {
      f32x4 a3[NT3];
#pragma unroll
      for (int nt = 0; nt < NT3; ++nt) a3[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NT2; ++kb)
#pragma unroll
        for (int nt = 0; nt < NT3; ++nt) {
          const f32x4 wq3 = w3l[(kb * NT3 + nt) * 64 + lane];
#if YL_BF16
          a3[nt] = YL_MFMA16(yl_pk_bf16(wq3), yl_pk_bf16(a2[kb]), a3[nt]);
#else
#pragma unroll
          for (int s = 0; s < 4; ++s)
            a3[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq3[s], a2[kb][s], a3[nt], 0, 0, 0);
#endif
        }
#pragma unroll
      for (int nt = 0; nt < NT3; ++nt) ov[nt] = clamp4(a3[nt] + bias3[nt], lo3, hi3);
    } else {
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) ov[nt] = a2[nt];
    }
  }
  flush_out();
}

// ------------------------------------------------------------------------------------------------
// yl_stemdw_kernel (round 6): the EfficientNet-Lite entry as ONE launch -- stem 3x3 s2 (3 -> 32, TF-SAME or symmetric pads)
// -> depthwise 3x3 s1 pad 1 (+BN+act) -> 1x1 (32 -> C3 <= 32, +BN): timm's conv_stem + bn1 and the DepthwiseSeparable block
// blocks.0.0 of tf_efficientnet_lite0..4 behind model_v2.py:94-100 (yololite_m: lite2).  As two launches (plain stem kernel,
// depthwise -> 1x1 kernel) the 32-channel stem output -- 13.1 MB per 640 x 640 image, the largest tensor of the network --
// is written and read back: 0.237 + 0.178 ms and 840 MB of traffic at B = 32.
// One WAVE owns an 8 x 8 output tile end to end (no workgroup barrier in the loop), like yl_stemblock_kernel:
//   stage   the [3 channels x 21 rows] x 21 input columns under the tile's 10 x 10 stem patch land in the wave's LDS region
//           by six LDS-DMA copies (interior tiles: uniform base + per-lane constant offsets; border tiles: dword-granular
//           with per-lane bounds, out-of-image elements read the zero buffer), requested right after the previous tile's
//           stem phase, so that they fly under its depthwise / 1x1 phases;
//   stem    seven 16-pixel m-tiles (100 patch pixels): 7 ds_read_b32 per lane and m-tile, 7 x 2 MFMAs (the bias rides in the
//           K = 27 -> 28 pad slot as in the stem block), clamp, zero outside the stem grid (the depthwise conv pads the STEM
//           OUTPUT) -> wave-private patch [10][10][32 + 4];
//   dw+pw   per output m-tile (two rows of eight pixels): the lane's 4 channels of both 16-channel blocks = bias + nine fma
//           (tap weights resident in registers) from the patch, clamp = B fragments of the 1x1 GEMM (2 x NT3 x 4 MFMAs, A
//           fragments resident in registers), + bias, clamp, one float4 NHWC store per lane.
// 1.56x the stem's MFMAs for the patch halo (100 stem pixels per 64 outputs) -- the launch is bound by its 0.37 GB of
// compulsory traffic and by VALU (288 fma per lane and tile), not by the matrix pipe.
#define SD_T 8
#define SD_P (SD_T + 2)
#define SD_NP (SD_P * SD_P)
#define SD_NM ((SD_NP + 15) / 16)
#define SD_IR (2 * (SD_P - 1) + 3)
#define SD_SP 24
#define SD_NSEG (3 * SD_IR)
#define SD_NCH (SD_NSEG * SD_SP / 4)
#define SD_NSTG ((SD_NCH + 63) / 64)
#define SD_NSTG1 ((SD_NSEG * SD_SP + 63) / 64)
template <int NT3 /*ceil(C3/16)*/, int NWV>
__global__ __launch_bounds__(NWV * 64, 2) void yl_stemdw_kernel(YlConvP p) {
  constexpr int NT1 = 2, C1 = 32, P1 = C1 + 4, KS = 7;
  // the stage holds exactly the 63 x 24 staged floats (the last copy is masked to its 58 live lanes): 20448 B per wave, so that
  // TWO 4-wave workgroups fit the 160 KiB of a CU (with whole 1 KiB copies: 2 x 82176 B = 512 B too many -- one wave per SIMD)
  constexpr int PATCH_F = SD_NP * P1, WAVE_F = PATCH_F + SD_NSEG * SD_SP;
  extern __shared__ __attribute__((aligned(16))) float sb_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  float* patch = sb_lds + wave * WAVE_F;
  float* stage = patch + PATCH_F;
  // ---- resident in registers: stem A fragments, depthwise taps + bias of the lane's channels, 1x1 A fragments + bias
  float wa[KS][NT1];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) wa[s][nt] = p.wp[(s * NT1 + nt) * 64 + lane];
#if YL_BF16
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) if (kq == 3) wa[6][nt] = 0.0f;
  yl_s16x4 wab[2][NT1];
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) {
    wab[0][nt] = yl_pk_bf16((f32x4){wa[0][nt], wa[1][nt], wa[2][nt], wa[3][nt]});
    wab[1][nt] = yl_pk_bf16((f32x4){wa[4][nt], wa[5][nt], wa[6][nt], 0.0f});
  }
  f32x4 bias1[NT1];
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) bias1[nt] = yl_ld4(p.bias + nt * 16 + 4 * kq);
  const bool slot6_lane = true;
#else
  const bool slot6_lane = kq != 3;                                   // lane group 3: the bias slot (input 1.0)
#endif
  f32x4 dww[9][NT1], dwb[NT1];
#pragma unroll
  for (int kb = 0; kb < NT1; ++kb) {
    dwb[kb] = yl_ld4(p.b2 + kb * 16 + 4 * kq);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) dww[tap][kb] = yl_ld4(p.w2p + tap * C1 + kb * 16 + 4 * kq);
  }
  f32x4 w3r[NT1][NT3], bias3[NT3];
#pragma unroll
  for (int nt = 0; nt < NT3; ++nt) {
    bias3[nt] = yl_ld4(p.b3 + nt * 16 + 4 * kq);
#pragma unroll
    for (int kb = 0; kb < NT1; ++kb) w3r[kb][nt] = reinterpret_cast<const f32x4*>(p.w3p)[(kb * NT3 + nt) * 64 + lane];
  }
  const int plane = p.H * p.W;
  int tky[KS], tkx[KS], tc[KS], ks[KS];                               // tap decode of the lane's k slots: see yl_stemblock_kernel
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    int row, kx;
    if (s < 3) { row = 2 * kq; kx = s; }
    else if (s < 6) { row = 2 * kq + 1; kx = s - 3; }
    else if (kq < 3) { row = 8; kx = kq; }
    else { row = 7; kx = 2; }
    tc[s] = row / 3; tky[s] = row - 3 * tc[s]; tkx[s] = kx;
    ks[s] = (tc[s] * SD_IR + tky[s]) * SD_SP + tkx[s];
  }
  int goff[SD_NSTG];
#pragma unroll
  for (int k = 0; k < SD_NSTG; ++k) {
    int e = k * 64 + lane;
    e = e < SD_NCH ? e : SD_NCH - 1;
    const int seg = e / (SD_SP / 4), ch = e - seg * (SD_SP / 4);
    const int c = seg / SD_IR, r = seg - c * SD_IR;
    goff[k] = c * plane + r * p.W + 4 * ch;
  }
  // the lane's patch pixel in each stem m-tile: stage base, patch write offset, patch coordinates
  int mbase[SD_NM], mdst[SD_NM], mrc[SD_NM];
#pragma unroll
  for (int m = 0; m < SD_NM; ++m) {
    const int q = 16 * m + pl, qq = q < SD_NP ? q : 0;
    const int r = qq / SD_P, c = qq - r * SD_P;
    mbase[m] = (2 * r) * SD_SP + 2 * c;
    mdst[m] = qq * P1 + 4 * kq;
    mrc[m] = (q < SD_NP ? 0 : (1 << 16)) | (r << 8) | c;
  }
  const float lo1 = (p.act == YL_ACT_NONE) ? -INFINITY : 0.0f, hi1 = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float lo2 = (p.act2 == YL_ACT_NONE) ? -INFINITY : 0.0f, hi2 = (p.act2 == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float lo3 = (p.act3 == YL_ACT_NONE) ? -INFINITY : 0.0f, hi3 = (p.act3 == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const int ty = pl >> 3, tx = pl & 7;                               // lane's pixel inside an output m-tile (2 rows x 8 columns)
  const int lr = (ty * SD_P + tx) * P1 + 4 * kq;                     // lane part of the patch read offset (+ (2 j + dy) rows, dx, kb)
  const int C3 = p.C3;
  const int tpr = (p.OW + SD_T - 1) / SD_T, tpc = (p.OH + SD_T - 1) / SD_T;
  const int tiles_img = tpr * tpc, ntiles = p.B * tiles_img;
  int r0, r1;                                                        // XCD bands (gridDim.x % 8 == 0), contiguous range per workgroup
  {
    const int gx = gridDim.x, bx = blockIdx.x;
    if ((gx & 7) == 0) {
      const int x = bx & 7, j = bx >> 3, nj = gx >> 3;
      const long b0 = ((long)ntiles * x) >> 3, b1 = ((long)ntiles * (x + 1)) >> 3;
      r0 = (int)(b0 + ((b1 - b0) * j) / nj); r1 = (int)(b0 + ((b1 - b0) * (j + 1)) / nj);
    } else {
      r0 = (int)(((long)ntiles * bx) / gx); r1 = (int)(((long)ntiles * (bx + 1)) / gx);
    }
  }
  const float* const xnet = reinterpret_cast<const float*>(p.x);     // the network input: fp32 NCHW in every build
  const float* const zf = reinterpret_cast<const float*>(p.zeros);
  auto gather = [&](int tile) {                                      // stage the input block under `tile`'s stem patch
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int tyi = trem / tpr, txi = trem - tyi * tpr;
    const int sy0 = tyi * SD_T - 1, sx0 = txi * SD_T - 1;
    const int iy0 = sy0 * p.stride - p.pad_t, ix0 = sx0 * p.stride - p.pad_l;
    const float* xb = xnet + (size_t)b * 3 * plane;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + SD_IR <= p.H && ix0 + SD_IR <= p.W &&
                          (ix0 + SD_SP <= p.W || iy0 + SD_IR < p.H || b + 1 < p.B);   // columns 21..23: inside the tensor
    if (interior) {
      const float* xo = xb + (long)iy0 * p.W + ix0;
#pragma unroll
      for (int k = 0; k < SD_NSTG; ++k)
        if (k + 1 < SD_NSTG || lane < SD_NCH - 64 * (SD_NSTG - 1)) yl_glds16(xo + goff[k], stage + k * 256);
    } else {
#pragma unroll
      for (int k = 0; k < SD_NSTG1; ++k) {
        int e = k * 64 + lane;
        e = e < SD_NSEG * SD_SP ? e : SD_NSEG * SD_SP - 1;
        const int seg = e / SD_SP, col = e - seg * SD_SP;
        const int c = seg / SD_IR, r = seg - c * SD_IR;
        const int iy = iy0 + r, ix = ix0 + col;
        const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        if (k + 1 < SD_NSTG1 || lane < SD_NSEG * SD_SP - 64 * (SD_NSTG1 - 1)) yl_glds4(in ? xb + c * plane + (long)iy * p.W + ix : zf, stage + k * 64);
      }
    }
  };
  int tile = r0 + wave;
  if (tile < r1) gather(tile);
  while (tile < r1) {
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int tyi = trem / tpr, txi = trem - tyi * tpr;
    const int sy0 = tyi * SD_T - 1, sx0 = txi * SD_T - 1;
    const bool inner = sy0 >= 0 && sx0 >= 0 && sy0 + SD_P <= p.SH && sx0 + SD_P <= p.SW;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the staged input has landed (and the last tile's stores left)
    // ---- stem on the 100 patch pixels -> wave-private patch.  The seven stage reads of m-tile m + 1 are requested before the
    // MFMAs of m-tile m (two register sets); two copies of the phase, selected by a wave-uniform branch: tiles whose patch lies
    // inside the stem grid carry no padding logic
    auto stem_phase = [&](auto inner_tag) {
      constexpr bool INNER = decltype(inner_tag)::value;
      float xs[2][KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) xs[0][s] = stage[mbase[0] + ks[s]];
#pragma unroll
      for (int m = 0; m < SD_NM; ++m) {
        if (m + 1 < SD_NM) {
#pragma unroll
          for (int s = 0; s < KS; ++s) xs[(m + 1) & 1][s] = stage[mbase[m + 1] + ks[s]];
        }
        const float (&x)[KS] = xs[m & 1];
        f32x4 a1[NT1];
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) a1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if YL_BF16
        {
          const yl_s16x4 x0 = yl_pk_bf16((f32x4){x[0], x[1], x[2], x[3]});
          const yl_s16x4 x1 = yl_pk_bf16((f32x4){x[4], x[5], x[6], 0.0f});
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt) {
            a1[nt] = YL_MFMA16(wab[0][nt], x0, a1[nt]);
            a1[nt] = YL_MFMA16(wab[1][nt], x1, a1[nt]);
          }
        }
#else
        const float x6 = slot6_lane ? x[6] : 1.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt)
            a1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][nt], s == 6 ? x6 : x[s], a1[nt], 0, 0, 0);
#endif
        const bool live = (16 * m + 15 < SD_NP) || (mrc[m] >> 16) == 0;          // only the last m-tile has dead lanes
        bool inside = true;
        if (!INNER) {
          const int sy = sy0 + ((mrc[m] >> 8) & 255), sx = sx0 + (mrc[m] & 255);
          inside = sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW;                  // else: the depthwise conv's zero padding
        }
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) {
#if YL_BF16
          f32x4 v = yl_clamp4(a1[nt] + bias1[nt], lo1, hi1);
#else
          f32x4 v = yl_clamp4(a1[nt], lo1, hi1);
#endif
          if (!INNER && !inside) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (live) *reinterpret_cast<f32x4*>(patch + mdst[m] + nt * 16) = v;
        }
      }
    };
    if (inner) stem_phase(std::true_type{});
    else stem_phase(std::false_type{});
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // patch writes of all lanes -> reads below; stage reads done
    __builtin_amdgcn_wave_barrier();
    const int next = tile + NWV;
    if (next < r1) gather(next);                                     // in flight under the depthwise / 1x1 phases
    // ---- depthwise 3x3 + 1x1 on the four output m-tiles
    const int oy0 = tyi * SD_T, ox0 = txi * SD_T;
#pragma unroll
    for (int j = 0; j < SD_T / 2; ++j) {
      f32x4 xq[NT1][1];
#pragma unroll
      for (int kb = 0; kb < NT1; ++kb) {
        f32x4 q = dwb[kb];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(patch + lr + ((2 * j + tap / 3) * SD_P + tap % 3) * P1 + kb * 16);
          q.x = fmaf(v.x, dww[tap][kb].x, q.x); q.y = fmaf(v.y, dww[tap][kb].y, q.y);
          q.z = fmaf(v.z, dww[tap][kb].z, q.z); q.w = fmaf(v.w, dww[tap][kb].w, q.w);
        }
        xq[kb][0] = yl_clamp4(q, lo2, hi2);
      }
      f32x4 a3[1][NT3];
#pragma unroll
      for (int nt = 0; nt < NT3; ++nt) a3[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NT1; ++kb) yl_mma_step<NT3, 1>(w3r[kb], xq[kb], a3);
      const int oy = oy0 + 2 * j + ty, ox = ox0 + tx;
      if (oy < p.OH && ox < p.OW) {
        yl_act_t* orow = p.out + (((size_t)b * p.OH + oy) * p.OW + ox) * C3;
#pragma unroll
        for (int nt = 0; nt < NT3; ++nt) {
          const int n = nt * 16 + 4 * kq;
          if (n < C3) yl_st4(orow + n, yl_clamp4(a3[0][nt] + bias3[nt], lo3, hi3));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // patch reads done before the next tile's writes
    __builtin_amdgcn_wave_barrier();
    tile = next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // no copy may land in LDS after the wave has ended
}

template <int NT3>
static hipError_t sd_go(const YlConvP& p, hipStream_t st, bool attr_only) {
  constexpr int NWV = 4;
  const size_t lds = (size_t)NWV * (SD_NP * 36 + SD_NSEG * SD_SP) * 4;                 // 81792 B: two workgroups per CU
  if (attr_only)
    return hipFuncSetAttribute((const void*)yl_stemdw_kernel<NT3, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const long ntiles = (long)p.B * ((p.OW + SD_T - 1) / SD_T) * ((p.OH + SD_T - 1) / SD_T);
  long gx = 2 * YL_NUM_CU;                                            // two 4-wave workgroups per CU (82 KB of LDS each)
  if (gx > (ntiles + NWV - 1) / NWV) gx = (ntiles + NWV - 1) / NWV;
  if (gx >= 8) gx &= ~7L;
  hipLaunchKernelGGL((yl_stemdw_kernel<NT3, NWV>), dim3((unsigned)gx), dim3(NWV * 64), lds, st, p);
  return hipGetLastError();
}

// stem 3x3 s2 (3 -> 32) -> depthwise 3x3 s1 pad 1 -> 1x1 (32 -> C3): p.w2p = depthwise taps [9][32], p.b2 [32], p.w3p / p.b3 the
// packed 1x1 (two k-blocks), p.SH x p.SW the stem grid = the output grid
hipError_t yl_launch_stemdw(const YlConvP& p, hipStream_t st) {
  if (p.stride != 2 || p.k != 3 || p.C1 != 32 || p.C2 != 32 || p.C3 < 4 || p.C3 > 32 || (p.C3 & 3) || p.OH != p.SH || p.OW != p.SW)
    return hipErrorInvalidValue;
  return p.C3 <= 16 ? sd_go<1>(p, st, false) : sd_go<2>(p, st, false);
}

template <int NT1, int NT2, int NT3, int NWV>
static hipError_t sb_launch(const YlConvP& p0, hipStream_t st, bool attr_only, size_t lds) {
  if (attr_only)
    return hipFuncSetAttribute((const void*)yl_stemblock_kernel<NT1, NT2, NT3, NWV>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  YlConvP p = p0;
  if (p.stride != 2 || p.k != 3) return hipErrorInvalidValue;   // the staged input block is 11 x 35 per channel
  // strips of ~10 tiles: long enough that the one extra m-tile of a strip's first tile is noise (1.4 of 147 MFMAs per
  // tile), short enough that every wave of the 8-waves-per-CU grid gets several (640 x 640, B = 64: 5 each).
  // (A makespan-minimising length -- 5 tiles for the 32-image chunks of the two-stream plan, 5 instead of 2.5 strips per
  // wave -- was measured SLOWER in the real step: 37.55k vs 37.8k images/s; the two chunk streams fill each other's
  // tails, fewer and longer strips cost less.)
  const int tpr = (p.OW + SB_TC - 1) / SB_TC, tpc = (p.OH + SB_TR - 1) / SB_TR;
  int nsp = (tpr + 5) / 10;
  if (nsp < 1) nsp = 1;
  // small batches (round 4): with ~10-tile strips one image is 160 strips = 20 workgroups walking 10 tiles each in series
  // (99 us at B = 1); shorter strips until ~2048 waves exist (one per wave slot of the 8-wave workgroups on 256 CUs).  A
  // tile's arithmetic does not depend on the strip it is part of: same bits.
  {
    const long rows = (long)p.B * tpc;
    const long want = (2048 + rows - 1) / rows;
    if (want > nsp) nsp = (int)(want < tpr ? want : tpr);
  }
  p.sb_strip = (tpr + nsp - 1) / nsp;
  const long nstrips = (long)p.B * tpc * ((tpr + p.sb_strip - 1) / p.sb_strip);
  int gx = (8 / NWV) * YL_NUM_CU;
  if (gx > (nstrips + NWV - 1) / NWV) gx = (int)((nstrips + NWV - 1) / NWV);
  if (gx >= 8) gx &= ~7;                                  // multiple of 8: XCD-aware strip ranges (see the kernel)
  hipLaunchKernelGGL((yl_stemblock_kernel<NT1, NT2, NT3, NWV>), dim3(gx), dim3(NWV * 64), lds, st, p);
  return hipGetLastError();
}

// One 8-wave workgroup per CU shares ONE copy of the conv2 / conv3 weights where that fits the 160 KB of LDS next to
// the eight wave-private patches and stages; otherwise 4 waves.
template <int NT1, int NT2, int NT3>
static hipError_t sb_go(const YlConvP& p0, hipStream_t st, bool attr_only) {
  constexpr int P1 = NT1 * 16 + 4;
  constexpr size_t wave_b = (size_t)(SB_NPATCH * P1 + SB_NSTG * 256) * 4;
  constexpr size_t w_b = (size_t)(9 * NT1 * NT2 + NT2 * NT3) * 1024;
  if constexpr (8 * wave_b + w_b <= 160 * 1024) return sb_launch<NT1, NT2, NT3, 8>(p0, st, attr_only, 8 * wave_b + w_b);
  else return sb_launch<NT1, NT2, NT3, 4>(p0, st, attr_only, 4 * wave_b + w_b);
}

template <int NT1>
static hipError_t sb_dispatch(const YlConvP& p, hipStream_t st, bool attr_only) {
  const int nt2 = (p.C2 + 15) / 16, nt3 = (p.C3 + 15) / 16;
  hipError_t e = hipSuccess;
  bool hit = false;
#define SB_CASE(A, B)                                                     \
  if (attr_only || (nt2 == A && nt3 == B)) {                              \
    hit = true;                                                           \
    if ((e = sb_go<NT1, A, B>(p, st, attr_only)) != hipSuccess) return e; \
  }
  SB_CASE(1, 0) SB_CASE(1, 1) SB_CASE(2, 0) SB_CASE(2, 2) SB_CASE(1, 2) SB_CASE(2, 1)
#undef SB_CASE
  return hit ? e : hipErrorInvalidValue;
}

hipError_t yl_stemblock_init() {
  YlConvP p{};
  hipError_t e = sb_dispatch<1>(p, nullptr, true);
  if (e != hipSuccess) return e;
  if ((e = sd_go<1>(p, nullptr, true)) != hipSuccess) return e;
  if ((e = sd_go<2>(p, nullptr, true)) != hipSuccess) return e;
  return sb_dispatch<2>(p, nullptr, true);
}

bool yl_stemblock_supported(int c1, int c2, int c3) {
  const int nt2 = (c2 + 15) / 16, nt3 = (c3 + 15) / 16;
  return (c1 == 16 || c1 == 32) && nt2 >= 1 && nt2 <= 2 && nt3 >= 0 && nt3 <= 2 && (c2 % 4 == 0) && (c3 % 4 == 0);
}

hipError_t yl_launch_stemblock(const YlConvP& p, hipStream_t st) {
  if (p.C1 == 32) return sb_dispatch<2>(p, st, false);
  if (p.C1 == 16) return sb_dispatch<1>(p, st, false);
  return hipErrorInvalidValue;
}
