// Head branch of a pyramid level under yl_predict as ONE launch:
//     depthwise 3x3 (+act)  ->  1x1 conv (+act)  ->  1x1 head-output conv  ->  decode
// (the reference's head: DWConvBlock trunk + 1x1 `out` conv, /root/reference/scripts/model/model_v2.py:24-41, 262-318;
// decode of utils_ms.py:26-123 in the epilogue as in yl_conv_pwt_kernel<.., DEC>).  Before: two launches per level --
// yl_conv_dwc_kernel (depthwise waves + GEMM waves, B fragments through LDS, one GEMM wave idle at 6 n-tiles) wrote
// the 96-channel trunk tensor, yl_conv_pwt_kernel read it back: 0.262 ms for the 80x80 level at B = 64, each half at
// ~45 % of its MFMA time.
//
// Here every wave is autonomous, as in the stem block: lane (kq, pl) = pixel pl of a 4x4 tile, 4 consecutive channels.
//   per 16-channel block kb:  nine float4 taps of the lane's pixel straight from L1/L2 (requested one block ahead),
//                             bias + 9 fma (tap order of yl_conv_dwc_kernel) + activation  -> B operand of block kb,
//                             6 n-tiles x 4 MFMAs against the trunk 1x1 weights (A fragments from LDS)
//   bias + clamp in registers: the D fragment of n-tile nt IS the B fragment of k-block nt of the next GEMM
//   6 k-blocks x NT3 n-tiles x 4 MFMAs against the head-output weights (LDS), decode epilogue (one wave = whole rows).
// Both weight images (standard A-fragment packs of the two layers, 36 KiB each at 96 channels) and the depthwise taps
// are copied to LDS once per workgroup (LDS-DMA); the grid is persistent, a workgroup walks a contiguous range of
// tiles with its 8 waves interleaved (8 neighbouring tiles in flight share their halo lines in L1 / the XCD's L2).
// Same k order, same arithmetic as the two kernels it replaces: bit-identical NMS inputs (tests/test_gpu_parity.py).
// Nothing is written but boxes / scores / classes: the trunk tensor never exists.
#include <stdlib.h>
#include "yl_internal.h"
#include "yl_dev.h"
#include "yl_epi.h"

#ifndef DPP_NW
#define DPP_NW 8                       // waves per workgroup
#endif
#ifndef DPP_WPE
#define DPP_WPE 2                      // waves per SIMD the register budget is set for
#endif

template <int KB /*Cin/16*/, int NT1 /*trunk n-tiles*/, int NT3 /*head-output n-tiles*/>
__global__ __launch_bounds__(DPP_NW * 64, DPP_WPE) void yl_conv_dpp_kernel(YlConvMulti mp) {
  int yl_k = 0;
  if (mp.n > 1 && (int)blockIdx.x >= mp.p[1].blk0) yl_k = 1;
  if (mp.n > 2 && (int)blockIdx.x >= mp.p[2].blk0) yl_k = 2;
  if (mp.n > 3 && (int)blockIdx.x >= mp.p[3].blk0) yl_k = 3;
  const YlConvP& p = mp.p[yl_k];
  const int bx = (int)blockIdx.x - p.blk0, gx = p.nblk;
  constexpr int Cin = KB * 16;
  extern __shared__ __attribute__((aligned(16))) float dpp_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  f32x4* w1l = reinterpret_cast<f32x4*>(dpp_lds);                    // [KB][NT1][64] float4
  f32x4* w3l = w1l + KB * NT1 * 64;                                  // [NT1][NT3][64] float4
  float* dwl = reinterpret_cast<float*>(w3l + NT1 * NT3 * 64);       // [9][Cin] taps, [Cin] depthwise bias
  float* b1l = dwl + 10 * Cin;                                       // [NT1 * 16] trunk bias (zero padded)
  float* b3l = b1l + NT1 * 16;                                       // [NT3 * 16] head-output bias (zero padded)
  {
    const f32x4* g1 = reinterpret_cast<const f32x4*>(p.wp);
    for (int r = wave; r < KB * NT1; r += DPP_NW) yl_glds16(g1 + r * 64 + lane, w1l + r * 64);
    const f32x4* g3 = reinterpret_cast<const f32x4*>(p.w3p);
    for (int r = wave; r < NT1 * NT3; r += DPP_NW) yl_glds16(g3 + r * 64 + lane, w3l + r * 64);
    yl_glds_floats(p.dw_w, dwl, 9 * Cin, tid, DPP_NW * 64);
    if (p.dw_b) yl_glds_floats(p.dw_b, dwl + 9 * Cin, Cin, tid, DPP_NW * 64);
    else for (int i = tid; i < Cin; i += DPP_NW * 64) dwl[9 * Cin + i] = 0.0f;
    yl_glds_floats(p.bias, b1l, NT1 * 16, tid, DPP_NW * 64);
    yl_glds_floats(p.b3, b3l, NT3 * 16, tid, DPP_NW * 64);
  }
  __syncthreads();

  const int H = p.H, W = p.W, OW = p.OW, OH = p.OH;                  // depthwise stride 1, 'same' padding: H == OH
  const int tw = OW >> 2, th = OH >> 2;
  const int tiles_img = tw * th;
  const int ntiles = p.B * tiles_img;
  // XCD-aware ranges (gx % 8 == 0 and blk0 % 8 == 0: workgroup b runs on XCD b % 8): XCD x owns the contiguous band
  // [x*T/8, (x+1)*T/8) of the tiles, its gx/8 workgroups split the band evenly
  int r0, r1;
  {
    const int x = bx & 7, j = bx >> 3, nj = gx >> 3;
    const long b0 = ((long)ntiles * x) >> 3, b1 = ((long)ntiles * (x + 1)) >> 3;
    r0 = (int)(b0 + ((b1 - b0) * j) / nj);
    r1 = (int)(b0 + ((b1 - b0) * (j + 1)) / nj);
  }
  const float* const xin = p.x;
  const float* const zl = p.zeros + 4 * kq;                          // >= 1 KiB of zeros: the same kb offsets apply
  const float lo1 = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi1 = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float dlo = (p.dw_act == YL_ACT_RELU || p.dw_act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float dhi = (p.dw_act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float* const tapw = dwl + 4 * kq;                            // tap t of block kb: tapw[t * Cin + kb * 16]

  // Software pipeline over the tap loads: the taps of block kb + 2 are requested while block kb is multiplied (two
  // register sets), and the first block of the NEXT tile is requested before this tile's second GEMM and decode -- the
  // trunk input comes from the Infinity Cache / HBM (it was written by the previous launch), ~1.5-2k cycles per
  // request against 768 cycles of MFMAs per block.  (One block ahead, nothing across tiles: 0.299 ms per B = 64
  // launch of the three levels; without the loads 0.255 ms.)
  YlPix px[1];
  const float* tp[9];
  auto setup = [&](int tile) {
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int tyi = trem / tw, txi = trem - tyi * tw;
    px[0].b = b; px[0].oy = 4 * tyi + (pl >> 2); px[0].ox = 4 * txi + (pl & 3); px[0].valid = true;
    px[0].lin = ((size_t)b * OH + px[0].oy) * OW + px[0].ox;
    // nine tap pointers of the lane's pixel (the zero buffer where a tap falls outside the image): centre pointer +
    // wave-uniform deltas, row / column validity computed once
    const float* const ctr = xin + (((long)b * H + px[0].oy) * W + px[0].ox) * Cin + 4 * kq;
    bool rok[3], cok[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int iy = px[0].oy - p.dw_pad_t + d, ix = px[0].ox - p.dw_pad_l + d;
      rok[d] = iy >= 0 && iy < H;
      cok[d] = ix >= 0 && ix < W;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - p.dw_pad_t, dx = tap % 3 - p.dw_pad_l;                // wave-uniform
      tp[tap] = (rok[tap / 3] && cok[tap % 3]) ? ctr + (dy * W + dx) * Cin : zl;
    }
  };
  f32x4 xa[9], xb[9];                                                // taps of the even / odd blocks
  auto fetch = [&](f32x4 (&dst)[9], int kb) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) dst[tap] = yl_ld4(tp[tap] + kb * 16);
  };
  int tile = r0 + wave;
  if (tile < r1) { setup(tile); fetch(xa, 0); }
  while (tile < r1) {
    const YlPix pxc = px[0];                                         // the tile computed now
    if (KB > 1) fetch(xb, 1);
    f32x4 acc1[1][NT1];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) acc1[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // depthwise result of block kb (bias + 9 fma in the tap order of yl_conv_dwc_kernel, activation) = B operand of kb
    auto dwise = [&](const f32x4 (&xt)[9], int kb) {
      f32x4 q = *reinterpret_cast<const f32x4*>(tapw + 9 * Cin + kb * 16);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        q = yl_fma4(xt[tap], *reinterpret_cast<const f32x4*>(tapw + tap * Cin + kb * 16), q);
      return yl_clamp4(q, dlo, dhi);                          // ReLU family only (SiLU is refused at launch): no branch in the region
    };
    // One region per block (the asm fences keep the scheduler from hoisting later blocks' 36 A-fragment reads: 256 VGPRs
    // + scratch): A fragments of block kb, the MFMAs of block kb, and -- independent of them, for the scheduler to
    // interleave -- the depthwise arithmetic of block kb + 1 and the tap requests of block kb + 2.
    f32x4 xq[1], xn;
    xq[0] = dwise(xa, 0);
    if (2 < KB) fetch(xa, 2);
    auto block = [&](f32x4 (&xnext)[9], int kb) {        // xnext: taps of block kb + 1 (then refilled with block kb + 3)
      f32x4 wq[NT1];
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) wq[nt] = w1l[(kb * NT1 + nt) * 64 + lane];
      if (kb + 1 < KB) {
        xn = dwise(xnext, kb + 1);
        if (kb + 3 < KB) fetch(xnext, kb + 3);
      }
      yl_mma_step<NT1, 1>(wq, xq, acc1);
      xq[0] = xn;
      asm volatile("" ::: "memory");
    };
#pragma unroll
    for (int kb = 0; kb < KB; kb += 2) {
      block(xb, kb);
      if (kb + 1 < KB) block(xa, kb + 1);
    }
    const int next = tile + DPP_NW;
    if (next < r1) { setup(next); fetch(xa, 0); }                    // in flight under the second GEMM and the decode
    // trunk epilogue in registers: D fragment of n-tile nt = B fragment of k-block nt of the head-output GEMM
    f32x4 acc3[1][NT3];
#pragma unroll
    for (int nt = 0; nt < NT3; ++nt) acc3[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NT1; ++kb) {
      f32x4 hq[1];
      hq[0] = yl_clamp4(acc1[0][kb] + *reinterpret_cast<const f32x4*>(b1l + kb * 16 + 4 * kq), lo1, hi1);
      f32x4 wq[NT3];
#pragma unroll
      for (int nt = 0; nt < NT3; ++nt) wq[nt] = w3l[(kb * NT3 + nt) * 64 + lane];
      yl_mma_step<NT3, 1>(wq, hq, acc3);
      asm volatile("" ::: "memory");
    }
    const YlPix pxd[1] = {pxc};
#pragma unroll
    for (int nt = 0; nt < NT3; ++nt) acc3[0][nt] += *reinterpret_cast<const f32x4*>(b3l + nt * 16 + 4 * kq);
    yl_epi_decode<NT3, 1, true, true>(p, acc3, pxd, 0, kq, lane);
    tile = next;
  }
}

// ------------------------------------------------------------------------------------------------
// yl_conv_dpw_kernel (round 6): the same chain, same arithmetic, with the depthwise INPUT WINDOW IN LDS.  The kernel above
// fetches nine float4 taps per lane and 16-channel block straight from L1/L2 in the MFMA lane layout (lane = channel
// group * 16 + pixel): the four lanes of a quad sit on four pixels = four cache lines, so every tap load keeps the CU's
// texture addresser busy for 64 cycles -- 8 waves x 9 loads x 64 cycles = 4608 addresser cycles per block round against
// 2 x 1536 MFMA cycles per SIMD: the launch is bound by the addresser (0.19 ms of its 0.27), not by the matrix pipe.
// Here (the recipe of yl_conv_wino2_kernel / yl_conv_dwl_kernel, kept wave-autonomous: no barrier in the loop):
//   window   the 6 x 6 input pixels under a wave's 4 x 4 tile, one 16-channel block at a time (64 B per pixel), land in a
//            wave-private LDS ring of NBUF buffers by THREE asynchronous LDS-DMA copies per block whose quads read the 64
//            contiguous bytes of ONE pixel (16 addresser cycles each; 18 copies per tile instead of 54 fragment loads;
//            out-of-image pixels read the zero buffer); the k-blocks of consecutive tiles form ONE stream: the copies of
//            block g + NBUF + 1 go out as soon as the taps of block g + 1 have been read, so the next tile's first blocks
//            are in flight under this tile's second GEMM and decode;
//   B        the lane's nine taps by ds_read_b128 from the window: slot = 4 P + (q ^ 2 (y & 1)), P = 6 y + x, q = channel
//            quad -- one LDS lane group of a read holds pixel rows {a, a + 3} with one channel quad and {a + 1, a + 2} with
//            its neighbour: the four pixels of a row segment differ in P mod 4, the two rows of a quad differ in y & 1, the
//            two quads in bit 0 -> 16 different 16-byte bank groups, conflict-free for every tap;
//   order    per block the 4 NT1 MFMAs run in groups of two n-tiles; in front of a group the wave issues the LDS reads it
//            needs next (the following group's A fragments, three taps + tap weights of the NEXT block's B), behind it the
//            fma chain of those taps -- no LDS round trip is waited for in front of an MFMA.
// Same tap order, fma chain, k order and epilogue as yl_conv_dpp_kernel: BIT-IDENTICAL to it ("dev_select" bit 16 keeps
// the tap-load kernel; tests/test_gpu_parity.py).
#ifndef DPW_NW
#define DPW_NW 8
#endif
#ifndef DPW_ABL
#define DPW_ABL 0                      // VARIANT BUILDS ONLY (tools/build_variant.sh ... -DDPW_ABL=<bits>; results WRONG, timing only):
#endif                                 // 1 no decode, 2 no second GEMM, 4 no first-GEMM MFMAs, 8 no window copies, 16 no tap reads / fma, 32 no per-tile setup
template <int KB /*Cin/16*/, int NT1 /*trunk n-tiles*/, int NT3 /*head-output n-tiles*/>
__global__ __launch_bounds__(DPW_NW * 64, DPW_NW / 4) void yl_conv_dpw_kernel(YlConvMulti mp) {
  int yl_k = 0;
  if (mp.n > 1 && (int)blockIdx.x >= mp.p[1].blk0) yl_k = 1;
  if (mp.n > 2 && (int)blockIdx.x >= mp.p[2].blk0) yl_k = 2;
  if (mp.n > 3 && (int)blockIdx.x >= mp.p[3].blk0) yl_k = 3;
  const YlConvP& p = mp.p[yl_k];
  const int bx = (int)blockIdx.x - p.blk0, gx = p.nblk;
  constexpr int Cin = KB * 16;
  constexpr int NBUF = (KB % 3 == 0) ? 3 : 2;                        // ring of window buffers (KB % NBUF == 0)
  constexpr int WSL = DPW_NW > 8 ? 144 : 192;                        // float4 slots per buffer: three copies, 144 used (12 waves: the third copy is 16 lanes wide)
  constexpr int G = NT1 / 2;                                         // MFMA groups of two n-tiles per block
  constexpr int TPG = (9 + G - 1) / G;                               // taps whose reads ride in front of one group
  static_assert(NT1 % 2 == 0 && KB % NBUF == 0 && NBUF + 1 <= KB, "yl_conv_dpw_kernel shape");
  extern __shared__ __attribute__((aligned(16))) float dpp_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  f32x4* w1l = reinterpret_cast<f32x4*>(dpp_lds);                    // [KB][NT1][64] float4
  f32x4* w3l = w1l + KB * NT1 * 64;                                  // [NT1][NT3][64] float4
  float* dwl = reinterpret_cast<float*>(w3l + NT1 * NT3 * 64);       // [9][Cin] taps, [Cin] depthwise bias
  float* b1l = dwl + 10 * Cin;                                       // [NT1 * 16] trunk bias (zero padded)
  float* b3l = b1l + NT1 * 16;                                       // [NT3 * 16] head-output bias (zero padded)
  f32x4* const winl = reinterpret_cast<f32x4*>(b3l + NT3 * 16) + wave * (NBUF * WSL);   // the wave's window ring
  {
    const f32x4* g1 = reinterpret_cast<const f32x4*>(p.wp);
    for (int r = wave; r < KB * NT1; r += DPW_NW) yl_glds16(g1 + r * 64 + lane, w1l + r * 64);
    const f32x4* g3 = reinterpret_cast<const f32x4*>(p.w3p);
    for (int r = wave; r < NT1 * NT3; r += DPW_NW) yl_glds16(g3 + r * 64 + lane, w3l + r * 64);
    yl_glds_floats(p.dw_w, dwl, 9 * Cin, tid, DPW_NW * 64);
    if (p.dw_b) yl_glds_floats(p.dw_b, dwl + 9 * Cin, Cin, tid, DPW_NW * 64);
    else for (int i = tid; i < Cin; i += DPW_NW * 64) dwl[9 * Cin + i] = 0.0f;
    yl_glds_floats(p.bias, b1l, NT1 * 16, tid, DPW_NW * 64);
    yl_glds_floats(p.b3, b3l, NT3 * 16, tid, DPW_NW * 64);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int H = p.H, W = p.W, OW = p.OW, OH = p.OH;                  // depthwise stride 1, pad 1: H == OH
  const int tw = OW >> 2, th = OH >> 2;
  const int tiles_img = tw * th;
  const int ntiles = p.B * tiles_img;
  int r0, r1;                                                        // XCD bands, see yl_conv_dpp_kernel
  {
    const int x = bx & 7, j = bx >> 3, nj = gx >> 3;
    const long b0 = ((long)ntiles * x) >> 3, b1 = ((long)ntiles * (x + 1)) >> 3;
    r0 = (int)(b0 + ((b1 - b0) * j) / nj);
    r1 = (int)(b0 + ((b1 - b0) * (j + 1)) / nj);
  }
  const float* const xin = p.x;
  const float lo1 = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi1 = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float dlo = (p.dw_act == YL_ACT_RELU || p.dw_act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float dhi = (p.dw_act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float* const tapw = dwl + 4 * kq;                            // tap t of block kb: tapw[t * Cin + kb * 16]
  // copy role of the lane in copy j: slot 64 j + lane = (pixel P = y * 6 + x of the window, stored quad)
  int cy[3], cx[3], cq[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int s = 64 * j + lane, P = s >> 2;
    cy[j] = P / 6; cx[j] = P - 6 * cy[j];
    cq[j] = s < 144 ? ((s & 3) ^ ((cy[j] & 1) << 1)) : -1;
  }
  // compute role: tap (dy, dx) of the lane's pixel (sy, sx) sits at slot tb[(sy + dy) & 1 ...] + 24 dy + 4 dx
  const int sy = pl >> 2, sx = pl & 3;
  const f32x4* const tb0 = winl + 4 * (6 * sy + sx) + (kq ^ ((sy & 1) << 1));          // rows sy, sy + 2
  const f32x4* const tb1 = winl + 4 * (6 * sy + sx) + (kq ^ (((sy + 1) & 1) << 1));    // row sy + 1
  auto tap_at = [&](int buf, int tap) -> f32x4 {
    const f32x4* const b = (tap / 3) == 1 ? tb1 : tb0;
    return b[buf * WSL + 24 * (tap / 3) + 4 * (tap % 3)];
  };

  // (Round 6 tried these copies through a raw buffer descriptor -- `buffer_load_dwordx4 ... offen lds`, scalar k-block offset, no
  // 64-bit add per request: +1.1 % on the headline -- and took it back: the ring below waits with PARTIAL counts (vmcnt(3 (NBUF - 1)):
  // "everything but the two youngest windows has landed"), which needs the copies to complete in issue order; with the buffer form
  // the FIRST launch on freshly allocated arenas lost a detection in 5 of 100 fresh processes (tools/flake_dbg.py; 0 of 100 with
  // global_load_lds, bisected to that commit).  The kernels that use the buffer form wait for vmcnt(0).)
  struct Src { const float* s[3]; };
  auto setup = [&](int tile, Src& src, YlPix& px) {
    const bool tv = tile < r1;
    const int tc = tv ? tile : r1 - 1;
    const int b = tc / tiles_img;
    const int trem = tc - b * tiles_img;
    const int tyi = trem / tw, txi = trem - tyi * tw;
    px.b = b; px.oy = 4 * tyi + sy; px.ox = 4 * txi + sx; px.valid = tv;
    px.lin = ((size_t)b * OH + px.oy) * OW + px.ox;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int gy = 4 * tyi - 1 + cy[j], gxx = 4 * txi - 1 + cx[j];
      const bool in = tv && cq[j] >= 0 && gy >= 0 && gy < H && gxx >= 0 && gxx < W;
      src.s[j] = in ? xin + (((long)b * H + gy) * W + gxx) * Cin + 4 * cq[j] : p.zeros;
    }
  };
  auto request = [&](const Src& src, int kb, int buf) {
    if (DPW_ABL & 8) return;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (j < 2 || WSL == 192 || lane < 16) yl_glds16(src.s[j] + kb * 16, winl + buf * WSL + j * 64);
  };

  Src cs, ns;
  YlPix pxc, pxn;
  int tile = r0 + wave;
  setup(tile, cs, pxc);
#pragma unroll
  for (int kb = 0; kb < NBUF; ++kb) request(cs, kb, kb);
  f32x4 xq[1];
  {
    __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * (NBUF - 1)));           // vmcnt(3 (NBUF - 1)): block 0 has landed
    f32x4 q = *reinterpret_cast<const f32x4*>(tapw + 9 * Cin);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) q = yl_fma4(tap_at(0, tap), *reinterpret_cast<const f32x4*>(tapw + tap * Cin), q);
    xq[0] = yl_clamp4(q, dlo, dhi);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    request(cs, NBUF, 0);
  }
  // A fragments: two register sets of two n-tiles that alternate from MFMA group to MFMA group (K * G groups per tile, an even
  // number: the set of a tile's first group is always set 0); the fragments of a group are read one group ahead, across
  // block and tile boundaries, so that no MFMA waits for the LDS read issued in front of it
  static_assert((KB * G) % 2 == 0, "A-fragment register sets alternate per group");
  f32x4 wq[2][2];
  wq[0][0] = w1l[0 * 64 + lane];
  wq[0][1] = w1l[1 * 64 + lane];
  while (tile < r1) {
    const int next = tile + DPW_NW;
    if (DPW_ABL & 32) { ns = cs; pxn = pxc; pxn.valid = next < r1; } else
    setup(next, ns, pxn);
    f32x4 acc1[1][NT1];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) acc1[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      // block kb is multiplied; B of block kb + 1 (the next tile's block 0 behind the last one) is built beside it
      const int kbn = (kb + 1) % KB, bufn = (kb + 1) % NBUF;
      __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * (NBUF - 1)));         // the window of block kb + 1 has landed
      f32x4 xn = *reinterpret_cast<const f32x4*>(tapw + 9 * Cin + kbn * 16);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int cur = (kb * G + g) & 1;
        f32x4 tx[TPG], tv[TPG];
        if (g + 1 < G) {
          wq[cur ^ 1][0] = w1l[(kb * NT1 + 2 * g + 2) * 64 + lane];
          wq[cur ^ 1][1] = w1l[(kb * NT1 + 2 * g + 3) * 64 + lane];
        } else if (kb + 1 < KB) {
          wq[cur ^ 1][0] = w1l[((kb + 1) * NT1 + 0) * 64 + lane];
          wq[cur ^ 1][1] = w1l[((kb + 1) * NT1 + 1) * 64 + lane];
        } else {                                                     // the head-output GEMM's first group
          wq[cur ^ 1][0] = w3l[0 * 64 + lane];
          wq[cur ^ 1][1] = w3l[1 * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < TPG; ++t)
          if (g * TPG + t < 9 && !(DPW_ABL & 16)) {
            tx[t] = tap_at(bufn, g * TPG + t);
            tv[t] = *reinterpret_cast<const f32x4*>(tapw + (g * TPG + t) * Cin + kbn * 16);
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < ((DPW_ABL & 4) ? 0 : 4); ++st) {
          acc1[0][2 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[cur][0][st], xq[0][st], acc1[0][2 * g], 0, 0, 0);
          acc1[0][2 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[cur][1][st], xq[0][st], acc1[0][2 * g + 1], 0, 0, 0);
        }
        if (DPW_ABL & 4) { acc1[0][2 * g] += wq[cur][0] * xq[0]; acc1[0][2 * g + 1] += wq[cur][1] * xq[0]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TPG; ++t)
          if (g * TPG + t < 9 && !(DPW_ABL & 16)) xn = yl_fma4(tx[t], tv[t], xn);
        __builtin_amdgcn_sched_barrier(0);
      }
      xq[0] = yl_clamp4(xn, dlo, dhi);
      // the buffer of block kb + 1 is free (its taps are in registers): it takes block kb + 1 + NBUF of the stream
      if (kb + 1 + NBUF < KB) request(cs, kb + 1 + NBUF, bufn);
      else request(ns, kb + 1 + NBUF - KB, bufn);
      __builtin_amdgcn_sched_barrier(0);
    }
    // trunk epilogue in registers: D fragment of n-tile nt = B fragment of k-block nt of the head-output GEMM
    f32x4 acc3[1][NT3];
#pragma unroll
    for (int nt = 0; nt < NT3; ++nt) acc3[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_assert(NT3 % 2 == 0 && (KB * G + NT1 * (NT3 / 2)) % 2 == 0, "A-fragment register sets alternate per group");
#pragma unroll
    for (int kb = 0; kb < NT1; ++kb) {
      f32x4 hq[1];
      hq[0] = yl_clamp4(acc1[0][kb] + *reinterpret_cast<const f32x4*>(b1l + kb * 16 + 4 * kq), lo1, hi1);
#pragma unroll
      for (int g = 0; g < NT3 / 2; ++g) {                            // groups of two n-tiles, A fragments one group ahead
        const int cur = (KB * G + kb * (NT3 / 2) + g) & 1;
        if (g + 1 < NT3 / 2) {
          wq[cur ^ 1][0] = w3l[(kb * NT3 + 2 * g + 2) * 64 + lane];
          wq[cur ^ 1][1] = w3l[(kb * NT3 + 2 * g + 3) * 64 + lane];
        } else if (kb + 1 < NT1) {
          wq[cur ^ 1][0] = w3l[((kb + 1) * NT3 + 0) * 64 + lane];
          wq[cur ^ 1][1] = w3l[((kb + 1) * NT3 + 1) * 64 + lane];
        } else {                                                     // the next tile's first group
          wq[cur ^ 1][0] = w1l[0 * 64 + lane];
          wq[cur ^ 1][1] = w1l[1 * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DPW_ABL & 2) { acc3[0][2 * g] += wq[cur][0] * hq[0]; acc3[0][2 * g + 1] += wq[cur][1] * hq[0]; } else
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          acc3[0][2 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[cur][0][st], hq[0][st], acc3[0][2 * g], 0, 0, 0);
          acc3[0][2 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[cur][1][st], hq[0][st], acc3[0][2 * g + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const YlPix pxd[1] = {pxc};
#pragma unroll
    for (int nt = 0; nt < NT3; ++nt) acc3[0][nt] += *reinterpret_cast<const f32x4*>(b3l + nt * 16 + 4 * kq);
    if (DPW_ABL & 1) {
      f32x4 t = acc3[0][0];
      for (int nt = 1; nt < NT3; ++nt) t += acc3[0][nt];
      if (kq == 0 && pxd[0].valid) p.dec_scores[(size_t)pxd[0].b * p.dec_N + p.dec_off + pxd[0].oy * p.OW + pxd[0].ox] = t.x + t.y + t.z + t.w;
    } else
    yl_epi_decode<NT3, 1, true, true>(p, acc3, pxd, 0, kq, lane);
    tile = next;
    cs = ns; pxc = pxn;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // no copy may land in LDS after the wave has ended
}

// ------------------------------------------------------------------------------------------------
// The same chain with a STORE epilogue and a wide middle: depthwise 3x3 (+act) -> 1x1 expand (+act) -> 1x1 project
// (+bias, +residual) -- MobileNetV4 UIB blocks with a start depthwise and no middle one (edge_n blocks.2.5: 48 -> 192 ->
// 48 at 40x40; timm `uir`, /root/reference/scripts/model/model_v2.py:79-121 consumes the features).  Before: two launches
// (yl_conv_dwh/dwt_kernel wrote the 192-channel tensor, yl_conv_pwt_kernel read it back: 79 MB each way at B = 64).
// The expanded row does not fit the registers at once: the expand GEMM runs in CHUNKS of 6 n-tiles (the depthwise
// results of all KB blocks stay in registers), and each chunk's 6 D fragments are consumed at once as 6 k-blocks of the
// project GEMM -- whose k order (ascending over all n-tiles of the expand) is that of the stand-alone launch.  The
// residual initialises the project accumulators as in yl_conv_pwt_kernel (pre-add).  Bit-identical to the two launches.
template <int KB /*Cin/16*/, int NC1 /*expand n-tiles / 6*/, int NT3 /*project n-tiles*/>
__global__ __launch_bounds__(DPP_NW * 64, DPP_WPE) void yl_conv_dpq_kernel(YlConvP p) {
  constexpr int Cin = KB * 16, NT1 = NC1 * 6;
  const int bx = (int)blockIdx.x, gx = (int)gridDim.x;
  extern __shared__ __attribute__((aligned(16))) float dpp_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  f32x4* w1l = reinterpret_cast<f32x4*>(dpp_lds);                    // [KB][NT1][64] float4
  f32x4* w3l = w1l + KB * NT1 * 64;                                  // [NT1][NT3][64] float4
  float* dwl = reinterpret_cast<float*>(w3l + NT1 * NT3 * 64);       // [9][Cin] taps, [Cin] depthwise bias
  float* b1l = dwl + 10 * Cin;                                       // [NT1 * 16] expand bias (zero padded)
  {
    const f32x4* g1 = reinterpret_cast<const f32x4*>(p.wp);
    for (int r = wave; r < KB * NT1; r += DPP_NW) yl_glds16(g1 + r * 64 + lane, w1l + r * 64);
    if constexpr (NT3 > 0) {
      const f32x4* g3 = reinterpret_cast<const f32x4*>(p.w3p);
      for (int r = wave; r < NT1 * NT3; r += DPP_NW) yl_glds16(g3 + r * 64 + lane, w3l + r * 64);
    }
    yl_glds_floats(p.dw_w, dwl, 9 * Cin, tid, DPP_NW * 64);
    if (p.dw_b) yl_glds_floats(p.dw_b, dwl + 9 * Cin, Cin, tid, DPP_NW * 64);
    else for (int i = tid; i < Cin; i += DPP_NW * 64) dwl[9 * Cin + i] = 0.0f;
    yl_glds_floats(p.bias, b1l, NT1 * 16, tid, DPP_NW * 64);
  }
  __syncthreads();
  const int H = p.H, W = p.W, OW = p.OW, OH = p.OH, N3 = p.C3;
  const int tw = OW >> 2, th = OH >> 2;
  const int tiles_img = tw * th;
  const int ntiles = p.B * tiles_img;
  int r0, r1;
  if ((gx & 7) == 0) {                                               // XCD bands, see yl_conv_dpp_kernel
    const int x = bx & 7, j = bx >> 3, nj = gx >> 3;
    const long b0 = ((long)ntiles * x) >> 3, b1 = ((long)ntiles * (x + 1)) >> 3;
    r0 = (int)(b0 + ((b1 - b0) * j) / nj);
    r1 = (int)(b0 + ((b1 - b0) * (j + 1)) / nj);
  } else {
    r0 = (int)(((long)ntiles * bx) / gx); r1 = (int)(((long)ntiles * (bx + 1)) / gx);
  }
  const float* const xin = p.x;
  const float* const zl = p.zeros + 4 * kq;
  const float lo1 = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi1 = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float lo3 = (p.act3 == YL_ACT_RELU || p.act3 == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi3 = (p.act3 == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float dlo = (p.dw_act == YL_ACT_RELU || p.dw_act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float dhi = (p.dw_act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float* const tapw = dwl + 4 * kq;
  const bool pre_add = p.res != nullptr && p.act3 == YL_ACT_NONE;

  size_t lin = 0;
  const float* tp[9];
  auto setup = [&](int tile) {
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int tyi = trem / tw, txi = trem - tyi * tw;
    const int oy = 4 * tyi + (pl >> 2), ox = 4 * txi + (pl & 3);
    lin = ((size_t)b * OH + oy) * OW + ox;
    const float* const ctr = xin + lin * Cin + 4 * kq;
    bool rok[3], cok[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int iy = oy - p.dw_pad_t + d, ix = ox - p.dw_pad_l + d;
      rok[d] = iy >= 0 && iy < H;
      cok[d] = ix >= 0 && ix < W;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - p.dw_pad_t, dx = tap % 3 - p.dw_pad_l;
      tp[tap] = (rok[tap / 3] && cok[tap % 3]) ? ctr + (dy * W + dx) * Cin : zl;
    }
  };
  f32x4 xa[9], xb[9];
  auto fetch = [&](f32x4 (&dst)[9], int kb) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) dst[tap] = yl_ld4(tp[tap] + kb * 16);
  };
  auto dwise = [&](const f32x4 (&xt)[9], int kb) {
    f32x4 q = *reinterpret_cast<const f32x4*>(tapw + 9 * Cin + kb * 16);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) q = yl_fma4(xt[tap], *reinterpret_cast<const f32x4*>(tapw + tap * Cin + kb * 16), q);
    return yl_clamp4(q, dlo, dhi);
  };
  int tile = r0 + wave;
  if (tile < r1) { setup(tile); fetch(xa, 0); }
  while (tile < r1) {
    const size_t linc = lin;                                         // the tile computed now
    if (KB > 1) fetch(xb, 1);
    f32x4 acc3[1][NT3 > 0 ? NT3 : 1];
#pragma unroll
    for (int nt = 0; nt < NT3; ++nt) {
      const int n = nt * 16 + 4 * kq;
      acc3[0][nt] = (pre_add && n < N3) ? yl_ld4(p.res + linc * N3 + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // depthwise results of every block stay in registers (xq); block kb + 1 is computed inside the region of the first
    // chunk's MFMAs of block kb, tap requests run two blocks ahead (as in yl_conv_dpp_kernel)
    f32x4 xq[KB][1];
    xq[0][0] = dwise(xa, 0);
    if (2 < KB) fetch(xa, 2);
    const int next = tile + DPP_NW;
#pragma unroll
    for (int c = 0; c < NC1; ++c) {
      f32x4 acc1[1][6];
#pragma unroll
      for (int nt = 0; nt < 6; ++nt) acc1[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        f32x4 wq[6];
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) wq[nt] = w1l[(kb * NT1 + c * 6 + nt) * 64 + lane];
        if (c == 0 && kb + 1 < KB) {
          if ((kb & 1) == 0) { xq[kb + 1][0] = dwise(xb, kb + 1); if (kb + 3 < KB) fetch(xb, kb + 3); }
          else { xq[kb + 1][0] = dwise(xa, kb + 1); if (kb + 3 < KB) fetch(xa, kb + 3); }
        }
        yl_mma_step<6, 1>(wq, xq[kb], acc1);
        asm volatile("" ::: "memory");                               // see yl_conv_dpp_kernel
      }
      if constexpr (NT3 == 0) {
        // wide expand only (no project conv): the chunk's six n-tiles are stored; the depthwise part is computed ONCE
        // per pixel (yl_conv_dwh_kernel recomputes it for every n-chunk of a layer with more than 96 outputs)
        if (c == NC1 - 1 && next < r1) { setup(next); fetch(xa, 0); }
        float* orow1 = p.out + linc * p.N;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int n = (c * 6 + j) * 16 + 4 * kq;
          *reinterpret_cast<f32x4*>(orow1 + n) =
              yl_clamp4(acc1[0][j] + *reinterpret_cast<const f32x4*>(b1l + (c * 6 + j) * 16 + 4 * kq), lo1, hi1);
        }
      } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        // next tile's first taps: in flight under the last 6 project steps and the epilogue.  (Requested earlier --
        // before the second chunk -- the kernel needs 256 VGPRs + scratch instead of 166.)
        if (c == NC1 - 1 && j == 0 && next < r1) { setup(next); fetch(xa, 0); }
        const int kb3 = c * 6 + j;
        f32x4 hq[1];
        hq[0] = yl_clamp4(acc1[0][j] + *reinterpret_cast<const f32x4*>(b1l + kb3 * 16 + 4 * kq), lo1, hi1);
        f32x4 wq[NT3 > 0 ? NT3 : 1];
#pragma unroll
        for (int nt = 0; nt < NT3; ++nt) wq[nt] = w3l[(kb3 * NT3 + nt) * 64 + lane];
        yl_mma_step<(NT3 > 0 ? NT3 : 1), 1>(wq, hq, acc3);
        asm volatile("" ::: "memory");
      }
      }
    }
    if constexpr (NT3 > 0) {
    float* orow = p.out + linc * N3;
#pragma unroll
    for (int nt = 0; nt < NT3; ++nt) {
      const int n = nt * 16 + 4 * kq;
      f32x4 v = yl_clamp4(acc3[0][nt] + yl_ld4(p.b3 + n), lo3, hi3);
      if (!pre_add && p.res && n < N3) v += yl_ld4(p.res + linc * N3 + n);
      if (n < N3) *reinterpret_cast<f32x4*>(orow + n) = v;
    }
    }
    tile = next;
  }
}

// shapes instantiated: (Cin/16, expand n-tiles / 6, project n-tiles)
#define YL_DPQ_SHAPES(X) X(3, 2, 3) X(3, 3, 0)

bool yl_dpq_supported(int cin, int cmid, int cout, int oh, int ow) {
  // cout == 0: the wide-expand-only form (no project conv)
  if ((cin & 15) || (cmid % 96) || (cout & 3) || (oh & 3) || (ow & 3)) return false;
#define YL_DPQ_CHECK(A, B, C) if (cin == A * 16 && cmid == B * 96 && (cout + 15) / 16 == C) return true;
  YL_DPQ_SHAPES(YL_DPQ_CHECK)
#undef YL_DPQ_CHECK
  return false;
}

template <int KB, int NC1, int NT3>
static hipError_t dpq_go(const YlConvP& p, hipStream_t st, bool attr_only) {
  const size_t lds = (size_t)(KB * NC1 * 6 + NC1 * 6 * NT3) * 1024 + (size_t)(10 * KB * 16 + NC1 * 96) * 4;
  if (attr_only)
    return hipFuncSetAttribute((const void*)yl_conv_dpq_kernel<KB, NC1, NT3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds);
  const long t = (long)p.B * (p.OH >> 2) * (p.OW >> 2);
  long nb = YL_NUM_CU * (DPP_WPE * 4 / DPP_NW);
  if (nb > (t + DPP_NW - 1) / DPP_NW) nb = (t + DPP_NW - 1) / DPP_NW;
  if (nb >= 8) nb &= ~7L;
  hipLaunchKernelGGL((yl_conv_dpq_kernel<KB, NC1, NT3>), dim3((unsigned)nb), dim3(DPP_NW * 64), lds, st, p);
  return hipGetLastError();
}

// p: the depthwise -> expand layer's parameters with w3p / b3 / C3 / act3, `res` and `out` of the project layer
hipError_t yl_launch_conv_dpq(const YlConvP& p, hipStream_t st) {
  const int c3 = p.w3p ? p.C3 : 0;                                   // no project conv: the wide-expand-only form
  if (p.k != 1 || p.dw_k != 3 || p.dw_stride != 1 || p.C1 > 0 || p.up || YL_SMOOTH(p.act) ||
      YL_SMOOTH(p.dw_act) || (p.w3p && YL_SMOOTH(p.act3)) || (!p.w3p && p.res) || p.dec_boxes ||
      p.H != p.OH || p.W != p.OW || !yl_dpq_supported(p.Cin, p.N, c3, p.OH, p.OW))
    return hipErrorNotSupported;
  const int kb = p.Cin / 16, nc1 = p.N / 96, nt3 = (c3 + 15) / 16;
#define YL_DPQ_RUN(A, B, C) if (kb == A && nc1 == B && nt3 == C) return dpq_go<A, B, C>(p, st, false);
  YL_DPQ_SHAPES(YL_DPQ_RUN)
#undef YL_DPQ_RUN
  return hipErrorNotSupported;
}

// ------------------------------------------------------------------------------------------------
// Dense 3x3 stride-2 conv on a 16-channel input with a 1x1 conv chained behind it (edge_n blocks.1.0 16 -> 48 @160 -> 80
// + blocks.1.1 48 -> 32, mobilenetv4_conv_small_050 `cn` blocks), wave-autonomous with the input patch STAGED in LDS like
// the stem block's: one wave owns a 2x8 output tile; the 5 x 17 input pixels under it are five contiguous 1088-byte rows
// of the NHWC tensor, copied into a wave-private LDS buffer by six asynchronous 16-byte-per-lane LDS-DMA loads (double
// buffered: the next tile's patch is requested before this tile is computed; out-of-image chunks read the zero buffer);
// per tap ONE ds_read_b128 is the B fragment (the linear patch layout costs bank conflicts on 9 reads per 132 MFMAs: not
// worth a padded copy through registers), A fragments of both convs from LDS (33 KiB shared by the 8 waves), the 1x1
// chained in registers, the tile's stores held back and issued in front of the next patch request (vmcnt is one
// in-order counter).  yl_conv_mfma_kernel<3,2,1> spent 4.5 VALU instructions per MFMA on per-tap address / bounds
// arithmetic (SQ counters: 1.5e7 VALU instructions x 4 cycles against 1.08e8 MFMA cycles per B = 64 launch) -- and fp32
// VALU time is MFMA time.  Same k order (tap-major), same epilogues: bit-identical.
template <int NT /*n-tiles of the 3x3*/, int NT3 /*n-tiles of the chained 1x1*/>
__global__ __launch_bounds__(DPP_NW * 64, 2) void yl_conv_s2c_kernel(YlConvP p) {
  constexpr int PR = 5, PC = 17, RCH = PC * 4, NCH = PR * RCH, NDMA = (NCH + 63) / 64;   // 340 16-byte chunks, 6 loads
  constexpr int BUF_F = NDMA * 256;                                  // floats per patch buffer
  const int bx = (int)blockIdx.x, gx = (int)gridDim.x;
  extern __shared__ __attribute__((aligned(16))) float dpp_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  f32x4* w2l = reinterpret_cast<f32x4*>(dpp_lds);                    // [9][NT][64] float4
  f32x4* w3l = w2l + 9 * NT * 64;                                    // [NT][NT3][64] float4
  float* b2l = reinterpret_cast<float*>(w3l + NT * NT3 * 64);        // [NT * 16], [NT3 * 16]
  float* b3l = b2l + NT * 16;
  float* patch = b3l + NT3 * 16 + wave * 2 * BUF_F;                  // two buffers per wave
  {
    const f32x4* g2 = reinterpret_cast<const f32x4*>(p.wp);
    for (int r = wave; r < 9 * NT; r += DPP_NW) yl_glds16(g2 + r * 64 + lane, w2l + r * 64);
    const f32x4* g3 = reinterpret_cast<const f32x4*>(p.w3p);
    for (int r = wave; r < NT * NT3; r += DPP_NW) yl_glds16(g3 + r * 64 + lane, w3l + r * 64);
    yl_glds_floats(p.bias, b2l, NT * 16, tid, DPP_NW * 64);
    yl_glds_floats(p.b3, b3l, NT3 * 16, tid, DPP_NW * 64);
  }
  __syncthreads();
  const int H = p.H, W = p.W, OW = p.OW, OH = p.OH, C3 = p.C3;
  const int tw = OW >> 3, th = OH >> 1;
  const int tiles_img = tw * th;
  const int ntiles = p.B * tiles_img;
  int r0, r1;
  if ((gx & 7) == 0) {                                               // XCD bands, see yl_conv_dpp_kernel
    const int x = bx & 7, j = bx >> 3, nj = gx >> 3;
    const long b0 = ((long)ntiles * x) >> 3, b1 = ((long)ntiles * (x + 1)) >> 3;
    r0 = (int)(b0 + ((b1 - b0) * j) / nj);
    r1 = (int)(b0 + ((b1 - b0) * (j + 1)) / nj);
  } else {
    r0 = (int)(((long)ntiles * bx) / gx); r1 = (int)(((long)ntiles * (bx + 1)) / gx);
  }
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float lo3 = (p.act3 == YL_ACT_RELU || p.act3 == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi3 = (p.act3 == YL_ACT_RELU6) ? 6.0f : INFINITY;
  // the lane's chunk of every DMA load: (patch row, pixel, 16-byte quarter of the pixel's 64 bytes)
  int crow[NDMA], cpx[NDMA], coff[NDMA];
#pragma unroll
  for (int k = 0; k < NDMA; ++k) {
    int e = k * 64 + lane;
    e = e < NCH ? e : NCH - 1;
    crow[k] = e / RCH;
    const int ch = e - crow[k] * RCH;
    cpx[k] = ch >> 2;
    coff[k] = (crow[k] * W + cpx[k]) * 16 + (ch & 3) * 4;
  }
  const int ty = pl >> 3, tx = pl & 7;
  const int rbase = ((2 * ty) * RCH + (2 * tx) * 4 + kq) * 4;       // tap (ky, kx): + (ky * RCH + kx * 4) * 4 floats
  auto request = [&](int tile, float* buf) {
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int tyi = trem / tw, txi = trem - tyi * tw;
    const int iy0 = 2 * (2 * tyi) - p.pad_t, ix0 = 2 * (8 * txi) - p.pad_l;
    const float* const img = p.x + (size_t)b * H * W * 16;
    if (iy0 >= 0 && ix0 >= 0 && iy0 + PR <= H && ix0 + PC <= W) {  // interior: uniform base + per-lane constants
      const float* const org = img + ((long)iy0 * W + ix0) * 16;
#pragma unroll
      for (int k = 0; k < NDMA; ++k) yl_glds16(org + coff[k], buf + k * 256);
    } else {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        const int iy = iy0 + crow[k], ix = ix0 + cpx[k];
        const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
        yl_glds16(in ? img + ((long)iy0 * W + ix0) * 16 + coff[k] : p.zeros, buf + k * 256);
      }
    }
  };
  f32x4 ov[NT3];
  float* o_row = p.out;
  bool o_valid = false;
  auto flush_out = [&]() {
#pragma unroll
    for (int j = 0; j < NT3; ++j) {
      const int n = j * 16 + 4 * kq;
      if (o_valid && n < C3) *reinterpret_cast<f32x4*>(o_row + n) = ov[j];
    }
  };
  int tile = r0 + wave, cur = 0;
  if (tile < r1) request(tile, patch);
  while (tile < r1) {
    const int next = tile + DPP_NW;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this tile's patch has landed
    flush_out();                                                     // previous tile's outputs, in front of the next request
    if (next < r1) request(next, patch + (cur ^ 1) * BUF_F);
    const float* pb = patch + cur * BUF_F + rbase;
    f32x4 acc[1][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      f32x4 xq[1], wq[NT];
      xq[0] = *reinterpret_cast<const f32x4*>(pb + ((tap / 3) * RCH + (tap % 3) * 4) * 4);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wq[nt] = w2l[(tap * NT + nt) * 64 + lane];
      yl_mma_step<NT, 1>(wq, xq, acc);
    }
    f32x4 a3[1][NT3];
#pragma unroll
    for (int j = 0; j < NT3; ++j) a3[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 v[1], w3[NT3];
      v[0] = yl_clamp4(acc[0][nt] + *reinterpret_cast<const f32x4*>(b2l + nt * 16 + 4 * kq), lo, hi);
#pragma unroll
      for (int j = 0; j < NT3; ++j) w3[j] = w3l[(nt * NT3 + j) * 64 + lane];
      yl_mma_step<NT3, 1>(w3, v, a3);
    }
    {
      const int b = tile / tiles_img;
      const int trem = tile - b * tiles_img;
      const int tyi = trem / tw, txi = trem - tyi * tw;
      const size_t lin = ((size_t)b * OH + 2 * tyi + ty) * OW + 8 * txi + tx;
      o_row = p.out + lin * C3;
      o_valid = true;
#pragma unroll
      for (int j = 0; j < NT3; ++j)
        ov[j] = yl_clamp4(a3[0][j] + *reinterpret_cast<const f32x4*>(b3l + j * 16 + 4 * kq), lo3, hi3);
    }
    tile = next;
    cur ^= 1;
  }
  flush_out();
}

static size_t s2c_lds_bytes(int nt, int nt3) {
  return (size_t)(9 * nt + nt * nt3) * 1024 + (size_t)(nt + nt3) * 64 + (size_t)DPP_NW * 2 * 6 * 1024;
}

bool yl_s2c_supported(int cin, int cout, int c3, int oh, int ow) {
  return cin == 16 && cout == 48 && c3 > 16 && c3 <= 32 && (c3 & 3) == 0 && (oh & 1) == 0 && (ow & 7) == 0;
}

// dense 3x3 stride-2 conv (16 -> 48) + chained 1x1 (-> 32).  hipErrorNotSupported: other shapes (yl_conv_mfma_kernel)
hipError_t yl_launch_conv_s2c(const YlConvP& p, hipStream_t st) {
  if ((p.dev & YL_DEV_S2C_OFF) || !p.w3p || p.k != 3 || p.stride != 2 || p.dw_k || p.C1 > 0 || p.res || p.up || p.in_shift || p.dec_boxes ||
      YL_SMOOTH(p.act) || YL_SMOOTH(p.act3) || !yl_s2c_supported(p.Cin, p.N, p.C3, p.OH, p.OW) ||
      p.OH != (p.H + 2 * p.pad_t - 3) / 2 + 1 || p.OW != (p.W + 2 * p.pad_l - 3) / 2 + 1)
    return hipErrorNotSupported;
  const long t = (long)p.B * (p.OH >> 1) * (p.OW >> 3);
  long nb = YL_NUM_CU;
  if (nb > (t + DPP_NW - 1) / DPP_NW) nb = (t + DPP_NW - 1) / DPP_NW;
  if (nb >= 8) nb &= ~7L;
  hipLaunchKernelGGL((yl_conv_s2c_kernel<3, 2>), dim3((unsigned)nb), dim3(DPP_NW * 64), s2c_lds_bytes(3, 2), st, p);
  return hipGetLastError();
}

static size_t dpp_lds_bytes(int kb, int nt1, int nt3) {
  return (size_t)(kb * nt1 + nt1 * nt3) * 1024 + (size_t)(10 * kb * 16 + nt1 * 16 + nt3 * 16) * 4;
}

// shapes instantiated: (Cin/16, trunk n-tiles, head-output n-tiles)
#define YL_DPP_SHAPES(X) X(6, 6, 6) X(4, 4, 6)

bool yl_dpp_supported(int cin, int cout, int c3, int oh, int ow) {
  if ((cin & 15) || (cout & 15) || (oh & 3) || (ow & 3)) return false;
#define YL_DPP_CHECK(A, B, C) if (cin == A * 16 && cout == B * 16 && (c3 + 15) / 16 == C) return true;
  YL_DPP_SHAPES(YL_DPP_CHECK)
#undef YL_DPP_CHECK
  return false;
}

template <int KB, int NT1, int NT3>
static hipError_t dpp_go(const YlConvP* ps, int n, hipStream_t st, bool attr_only) {
  const size_t lds = dpp_lds_bytes(KB, NT1, NT3);
  const size_t ldsw = lds + (size_t)DPW_NW * ((KB % 3 == 0) ? 3 : 2) * (DPW_NW > 8 ? 144 : 192) * 16;   // + the waves' window rings
  if (attr_only) {
    const hipError_t e = hipFuncSetAttribute((const void*)yl_conv_dpw_kernel<KB, NT1, NT3>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)yl_conv_dpp_kernel<KB, NT1, NT3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds);
  }
  // window-in-LDS form (round 6): depthwise pad 1 on every side; "dev_select" bit 16 keeps the tap-load kernel
  bool win = !(ps[0].dev & YL_DEV_DPW_OFF);
  for (int k = 0; k < n; ++k) win = win && ps[k].dw_pad_t == 1 && ps[k].dw_pad_l == 1;
  const int nw = win ? DPW_NW : DPP_NW;
  YlConvMulti m;
  m.n = n;
  long total = 0;
  for (int k = 0; k < n; ++k) { m.p[k] = ps[k]; total += (long)ps[k].B * (ps[k].OH >> 2) * (ps[k].OW >> 2); }
  // one workgroup per CU, dealt to the problems in proportion to their tiles in multiples of 8 (XCD-aligned ranges); a
  // launch smaller than that gets one workgroup per `nw` tiles
  const int budget = YL_NUM_CU;
  int at = 0;
  for (int k = 0; k < n; ++k) {
    const long t = (long)ps[k].B * (ps[k].OH >> 2) * (ps[k].OW >> 2);
    long nb = (t * budget + total / 2) / total;
    if (nb > (t + nw - 1) / nw) nb = (t + nw - 1) / nw;
    nb = (nb + 4) / 8 * 8;
    if (nb < 8) nb = 8;
    m.p[k].blk0 = at;
    m.p[k].nblk = (int)nb;
    at += (int)nb;
  }
  if (win) hipLaunchKernelGGL((yl_conv_dpw_kernel<KB, NT1, NT3>), dim3((unsigned)at), dim3(DPW_NW * 64), ldsw, st, m);
  else hipLaunchKernelGGL((yl_conv_dpp_kernel<KB, NT1, NT3>), dim3((unsigned)at), dim3(DPP_NW * 64), lds, st, m);
  return hipGetLastError();
}

// n <= 4 head branches of identical configuration (ps[k]: the trunk layer's parameters with w3p / b3 / C3 and the dec_*
// fields of its head-output layer).  hipErrorNotSupported: shape not instantiated (the two-launch form runs).
hipError_t yl_launch_conv_dpp(const YlConvP* ps, int n, hipStream_t st) {
  if (n < 1 || n > 4) return hipErrorNotSupported;
  const YlConvP& q = ps[0];
  if (q.k != 1 || q.dw_k != 3 || q.dw_stride != 1 || q.C1 > 0 || q.res || q.up || !q.w3p || !q.dec_boxes || q.dec_raw ||
      YL_SMOOTH(q.act) || YL_SMOOTH(q.dw_act))
    return hipErrorNotSupported;
  for (int k = 0; k < n; ++k)
    if (!yl_dpp_supported(ps[k].Cin, ps[k].N, ps[k].C3, ps[k].OH, ps[k].OW) || ps[k].Cin != q.Cin || ps[k].N != q.N ||
        ps[k].C3 != q.C3 || ps[k].H != ps[k].OH || ps[k].W != ps[k].OW ||
        (size_t)ps[k].B * ps[k].H * ps[k].W * ps[k].Cin * 4 >= ((size_t)1 << 31))        // (32-bit byte offsets of the window copies)
      return hipErrorNotSupported;
  const int kb = q.Cin / 16, nt1 = q.N / 16, nt3 = (q.C3 + 15) / 16;
#define YL_DPP_RUN(A, B, C) if (kb == A && nt1 == B && nt3 == C) return dpp_go<A, B, C>(ps, n, st, false);
  YL_DPP_SHAPES(YL_DPP_RUN)
#undef YL_DPP_RUN
  return hipErrorNotSupported;
}

// ------------------------------------------------------------------------------------------------
// Dense 3x3 (stride 1, pad 1) with FEW channels on both sides -- 16 or 32 in, <= 16 out -- on large grids (round 6): the first
// fused-MBConv blocks of efficientnetv2 (timm `er` / `cn` blocks behind model_v2.py:94-100: 32 -> 16 and 16 -> 16 at 320 x 320).
// They ran through yl_conv_wino_kernel with one of its two n-tiles empty (0.48 / 0.34 ms at B = 32: 206 vector instructions per
// 128 MFMAs of which 64 are useful) or, without Winograd, through yl_conv_mfma_kernel's nine fragment-shaped tap loads per
// k-block.  Here, with the machinery of yl_conv_dpw_kernel: one WAVE = one 4 x 4-pixel m-tile; its 6 x 6-pixel input window of
// every 16-channel block lands in a wave-private LDS region by buffer-descriptor LDS-DMA copies (quads on the 64 contiguous bytes
// of one pixel; pixels outside the image: out-of-range offset = zeros), the 9 KB taps are conflict-free ds_read_b128 in fragment
// order, ALL weight fragments (9 taps x KB blocks, one n-tile) live in registers for the whole launch, and the next tile's
// windows are requested as soon as this tile's taps are in registers -- they fly under its 36 KB MFMAs.  No workgroup barrier, no
// vector instruction computes an address.  Same k order (tap-major, k-blocks inside), pre-add rule and epilogues as
// yl_conv_mfma_kernel: BIT-IDENTICAL to the direct path ("dev_select" bit 17 switches the kernel off; tests/test_gpu_parity.py).
#define K3W_NW 8
template <int KB /*Cin/16*/>
__global__ __launch_bounds__(K3W_NW * 64, 2) void yl_conv_k3w_kernel(YlConvP p) {
  constexpr int Cin = KB * 16, WSL = 192;
  extern __shared__ __attribute__((aligned(16))) float dpp_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  f32x4* const winl = reinterpret_cast<f32x4*>(dpp_lds) + wave * (KB * WSL);      // the wave's windows: [KB][WSL] float4
  f32x4 wr[9][KB];                                                   // A fragments of (tap, k-block): pack_conv [tap][kb][1 n-tile][64]
  {
    const f32x4* const wg = reinterpret_cast<const f32x4*>(p.wp);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) wr[tap][kb] = wg[(tap * KB + kb) * 64 + lane];
  }
  const int H = p.H, W = p.W, OW = p.OW, OH = p.OH, N = p.N;
  const int tw = OW >> 2, th = OH >> 2;
  const int tiles_img = tw * th;
  const int ntiles = p.B * tiles_img;
  const int bx = blockIdx.x, gx = gridDim.x;
  int r0, r1;                                                        // XCD bands, see yl_conv_dpp_kernel
  if ((gx & 7) == 0) {
    const int x = bx & 7, j = bx >> 3, nj = gx >> 3;
    const long b0 = ((long)ntiles * x) >> 3, b1 = ((long)ntiles * (x + 1)) >> 3;
    r0 = (int)(b0 + ((b1 - b0) * j) / nj);
    r1 = (int)(b0 + ((b1 - b0) * (j + 1)) / nj);
  } else {
    r0 = (int)(((long)ntiles * bx) / gx); r1 = (int)(((long)ntiles * (bx + 1)) / gx);
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)((long)p.B * H * W * Cin * 4), 0x00020000);
  const bool pre_add = p.res != nullptr && p.act == YL_ACT_NONE;     // (as yl_conv_mfma_kernel: the addend initialises the accumulator)
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  // copy role of the lane in copy j: slot 64 j + lane = (pixel P = y * 6 + x of the window, stored quad) -- yl_conv_dpw_kernel's layout
  int cy[3], cx[3], cq[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int s = 64 * j + lane, P = s >> 2;
    cy[j] = P / 6; cx[j] = P - 6 * cy[j];
    cq[j] = s < 144 ? ((s & 3) ^ ((cy[j] & 1) << 1)) : -1;
  }
  const int sy = pl >> 2, sx = pl & 3;
  const f32x4* const tb0 = winl + 4 * (6 * sy + sx) + (kq ^ ((sy & 1) << 1));          // rows sy, sy + 2
  const f32x4* const tb1 = winl + 4 * (6 * sy + sx) + (kq ^ (((sy + 1) & 1) << 1));    // row sy + 1
  auto tap_at = [&](int kb, int tap) -> f32x4 {
    const f32x4* const b = (tap / 3) == 1 ? tb1 : tb0;
    return b[kb * WSL + 24 * (tap / 3) + 4 * (tap % 3)];
  };
  struct Src { unsigned s[3]; };
  auto setup = [&](int tile, Src& src, YlPix& px) {
    const bool tv = tile < r1;
    const int tc = tv ? tile : r1 - 1;
    const int b = tc / tiles_img;
    const int trem = tc - b * tiles_img;
    const int tyi = trem / tw, txi = trem - tyi * tw;
    px.b = b; px.oy = 4 * tyi + sy; px.ox = 4 * txi + sx; px.valid = tv;
    px.lin = ((size_t)b * OH + px.oy) * OW + px.ox;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int gy = 4 * tyi - 1 + cy[j], gxx = 4 * txi - 1 + cx[j];
      const bool in = tv && cq[j] >= 0 && gy >= 0 && gy < H && gxx >= 0 && gxx < W;
      src.s[j] = in ? (unsigned)((((b * H + gy) * W + gxx) * Cin + 4 * cq[j]) * 4) : 0x80000000u;
    }
  };
  auto request = [&](const Src& src) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        // (all 64 lanes, also in the third copy whose lanes >= 16 carry the out-of-range offset and write zeros into unused slots:
        //  an exec-masked LDS-DMA copy issued at the start of the kernel left the first window of every wave wrong)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(winl + kb * WSL + j * 64), 16, (int)src.s[j], kb * 64, 0, 0);
  };
  Src cs, ns;
  YlPix pxc, pxn;
  int tile = r0 + wave;
  setup(tile, cs, pxc);
  if (tile < r1) request(cs);
  while (tile < r1) {
    const int next = tile + K3W_NW;
    setup(next, ns, pxn);
    f32x4 acc[1][1];
    acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pre_add) {
      const int n = 4 * kq;
      if (n < N) acc[0][0] = yl_ld4(p.res + pxc.lin * N + n);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0): this tile's windows (and the residual row) have landed;
    asm volatile("" ::: "memory");                                   // the builtin keeps the compiler's counter model right, the asm keeps
    f32x4 xt[KB][9];                                                 // the tap reads below it
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) xt[kb][tap] = tap_at(kb, tap);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the taps are in registers: the windows are free
    if (next < r1) request(ns);                                      // the next tile's windows fly under this tile's MFMAs
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int st = 0; st < 4; ++st)
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[tap][kb][st], xt[kb][tap][st], acc[0][0], 0, 0, 0);
    const YlPix pxd[1] = {pxc};
    if (!pre_add && (p.res || YL_SMOOTH(p.act))) yl_epi_generic<1, 1>(p, acc, pxd, 0, kq);
    else yl_epi_fast<1, 1>(p, acc, pxd, 0, kq, lo, hi, true);
    tile = next;
    cs = ns; pxc = pxn;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // no copy may land in LDS after the wave has ended
}

template <int KB>
static hipError_t k3w_go(const YlConvP& p, hipStream_t st) {
  const size_t lds = (size_t)K3W_NW * KB * 192 * 16;
  const long t = (long)p.B * (p.OH >> 2) * (p.OW >> 2);
  static int res = 0;
  if (!res) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)yl_conv_k3w_kernel<KB>, K3W_NW * 64, lds) != hipSuccess || nb < 1) nb = 1;
    if (nb > 2) nb = 2;
    res = nb * YL_NUM_CU;
  }
  long nb = res;
  if (nb > (t + K3W_NW - 1) / K3W_NW) nb = (t + K3W_NW - 1) / K3W_NW;
  if (nb >= 8) nb &= ~7L;
  hipLaunchKernelGGL((yl_conv_k3w_kernel<KB>), dim3((unsigned)nb), dim3(K3W_NW * 64), lds, st, p);
  return hipGetLastError();
}

// dense 3x3 stride-1 pad-1 layers with 16 / 32 input and <= 16 output channels (one n-tile), 4x4-tileable grids of >= 160 x 160 pixels; fp32 storage and arithmetic only (the caller keeps the reduced-precision units on their own kernels)
hipError_t yl_launch_conv_k3w(const YlConvP& p, hipStream_t st) {
  if (p.k != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.dw_k > 0 || p.C1 > 0 || p.w3p || p.up || p.dec_boxes || p.scale ||
      p.in_shift || p.ldo || (p.N & 3) || p.NTtot != 1 || (p.Cin != 16 && p.Cin != 32) || (p.OH & 3) || (p.OW & 3) || p.H != p.OH || p.W != p.OW ||
      (p.dev & YL_DEV_K3W_OFF))
    return hipErrorNotSupported;
  // Only where the DIRECT kernel would run (option "winograd" 0, or layers without a Winograd image): there the results are the same
  // bits, so nothing about parity moves.  Replacing the first-form Winograd kernel on these layers under the default options is
  // worth +2.4 % on efficientnetv2 yololite_m (2.28 -> 2.34 k images/s) but changes low-order bits, and in the full-size parity
  // sample one box coordinate then sits 1.2e-4 px from the oracle's on a .5 rounding boundary (bar: 1e-4): not the default.
  if (p.wino) return hipErrorNotSupported;
  if ((p.OH >> 2) * (p.OW >> 2) < 1600) return hipErrorNotSupported;    // grids below 160 x 160: the other kernels (by SHAPE, not by batch: batch-invariant results)
  if ((size_t)p.B * p.H * p.W * p.Cin * 4 >= ((size_t)1 << 31)) return hipErrorNotSupported;            // 32-bit byte offsets
  return p.Cin == 16 ? k3w_go<1>(p, st) : k3w_go<2>(p, st);
}

hipError_t yl_dpp_init() {
  hipError_t e = hipFuncSetAttribute((const void*)yl_conv_s2c_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)s2c_lds_bytes(3, 2));
  YlConvP q0{};
#define YL_DPQ_ATTR(A, B, C) if (e == hipSuccess) e = dpq_go<A, B, C>(q0, nullptr, true);
  YL_DPQ_SHAPES(YL_DPQ_ATTR)
#undef YL_DPQ_ATTR
#define YL_DPP_ATTR(A, B, C) if (e == hipSuccess) e = dpp_go<A, B, C>(nullptr, 0, nullptr, true);
  YL_DPP_SHAPES(YL_DPP_ATTR)
#undef YL_DPP_ATTR
  return e;
}
