// Block-cooperative depthwise -> 1x1 convolution for gfx950 (MI355X / CDNA4), fp32 in / fp32 accumulate.
//
// Same layer as yl_conv_dwh_kernel (yl_conv.hip): depthwise DK x DK (stride DS) + bias + act, then a 1x1 conv
// (dw -> pw pairs of the reference's DWConvBlock, model_v2.py:23-39, and of the timm UIB / inverted-residual
// blocks behind model_v2.py:94-100,266-272), the depthwise result never touching HBM.  Different decomposition,
// built for the layers whose pixel count cannot fill the chip (40x40 / 20x20 grids: a few hundred to a few
// thousand 16-pixel tiles per launch).  yl_conv_dwh_kernel gives every WAVE a 4x4-pixel tile end to end: the wave
// walks all K/16 channel blocks serially (halo fetch -> taps -> MFMAs for every n-tile), one wave per SIMD, and
// every block first copies the whole 1x1 weight matrix into LDS.  rocprofv3 (profiles/r01_pmc_summary.txt): such
// a wave lives 52 k cycles of which 7.5 k are MFMA issue -- the rest is waiting for memory with nothing else
// resident to run.
//
// Here the FOUR waves of a workgroup share one 4x4-pixel tile:
//   phase 1 (K split)  wave w takes channel blocks w, w+4, ...: stages the (3*DS+DK)^2 halo patch of 16 channels in
//                      its private LDS region (loads issued one block ahead, also across tiles), forms
//                      B = act(bias + sum_taps w*x) and writes the fragment to a block-shared LDS buffer;
//   barrier
//   phase 2 (N split)  wave w owns n-tiles [w*NTW, (w+1)*NTW): its 1x1 weights (MFMA A fragments, all K) were
//                      loaded ONCE per workgroup into REGISTERS -- no LDS weight image, no weight prologue per
//                      tile -- and it runs the k loop over the shared B fragments, then the epilogue.
// The serial chain per tile is 4x shorter, 4x as many waves are resident per tile, and because every output
// still sums its k blocks in ascending order the results are BIT-IDENTICAL to yl_conv_dwh_kernel / the generic
// conv kernel (tests/test_gpu_parity.py: test_convc_kernels_are_bitwise_the_kernels_they_replace).
//
// Limits (launcher falls back to yl_conv_dwh_kernel otherwise): OH, OW multiples of 4, N % 4 == 0,
// NTW = ceil(ceil(N/16)/4) <= 5 and ceil(Cin/16) <= KBMAX(NTW) (the register budget of the resident weights).
#include "yl_lp.h"
#if defined(YL_BF16) && YL_BF16
#define yl_conv_dwc_kernel YL_LP_NAME(yl_conv_dwc_kernel)
#define yl_launch_conv_dwc YL_LP_NAME(yl_launch_conv_dwc)
#define yl_convc_init YL_LP_NAME(yl_convc_init)
#define yl_conv_pwt_kernel YL_LP_NAME(yl_conv_pwt_kernel)
#define yl_launch_conv_pwt YL_LP_NAME(yl_launch_conv_pwt)
#define yl_launch_conv_pwt_multi YL_LP_NAME(yl_launch_conv_pwt_multi)
#define yl_conv_dwt_kernel YL_LP_NAME(yl_conv_dwt_kernel)
#define yl_launch_conv_dwt YL_LP_NAME(yl_launch_conv_dwt)
#define yl_conv_dwk_kernel YL_LP_NAME(yl_conv_dwk_kernel)
#define yl_conv_dwl_kernel YL_LP_NAME(yl_conv_dwl_kernel)
#define yl_launch_conv_dwk YL_LP_NAME(yl_launch_conv_dwk)
#define yl_conv_dws_kernel YL_LP_NAME(yl_conv_dws_kernel)
#define yl_launch_conv_dws YL_LP_NAME(yl_launch_conv_dws)
#define yl_conv_wino_kernel YL_LP_NAME(yl_conv_wino_kernel)
#define yl_launch_conv_wino YL_LP_NAME(yl_launch_conv_wino)
#define yl_conv_kxk_kernel YL_LP_NAME(yl_conv_kxk_kernel)
#define yl_launch_conv_kxk YL_LP_NAME(yl_launch_conv_kxk)
#define yl_conv_pws_kernel YL_LP_NAME(yl_conv_pws_kernel)
#define yl_launch_conv_pws YL_LP_NAME(yl_launch_conv_pws)
#define yl_ir_kernel YL_LP_NAME(yl_ir_kernel)
#define yl_launch_conv_ir YL_LP_NAME(yl_launch_conv_ir)
#define yl_ir_supported YL_LP_NAME(yl_ir_supported)
#endif
#include <stdlib.h>
#include <map>
#include <type_traits>
#include <mutex>
#include <utility>
#include "yl_internal.h"
#include "yl_dev.h"
#include "yl_epi.h"
#if defined(YL_F16S) && YL_F16S
#define defined_YL_F16S 1
#else
#define defined_YL_F16S 0
#endif

// co-resident workgroups of `kernel` on the whole device (occupancy query cached per kernel and LDS size)
template <typename K>
static int yl_resident_blocks_n(K kernel, int threads, size_t lds) {
  static std::mutex mu;
  static std::map<std::pair<const void*, size_t>, int> cache;
  const std::pair<const void*, size_t> key((const void*)kernel, lds);
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kernel, threads, lds) != hipSuccess || nb < 1) nb = 1;
  if (nb > 8) nb = 8;
  cache[key] = nb * YL_NUM_CU;
  return nb * YL_NUM_CU;
}

#define YL_DWC_LDS_MAX (150 * 1024)

// profiling aid (variant builds only: tools/build_variant.sh stamp yl_convc.hip -DYL_DWC_STAMP=<Cin>): shader-clock
// stamps of the launches whose Cin matches, [block][wave][32], read back by tools/dwc_stamps.py
#ifdef YL_DWC_STAMP
__device__ unsigned long long yl_dwc_stamps[1024 * 8 * 32];
#define DWC_STAMP(i)                                                                                             \
  do {                                                                                                           \
    if (p.Cin == YL_DWC_STAMP && lane == 0 && blockIdx.x < 1024 && (i) < 32)                                     \
      yl_dwc_stamps[((size_t)blockIdx.x * 8 + wave) * 32 + (i)] = __builtin_readcyclecounter();                 \
  } while (0)
extern "C" int yl_debug_dwc_stamps(void* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(yl_dwc_stamps), sizeof(yl_dwc_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define DWC_STAMP(i) do {} while (0)
#endif
// the same aid for yl_conv_wino2_kernel / yl_conv_dwl_kernel / yl_ir_kernel (-DYL_WINO_STAMP=<Cin> [-DYL_STAMP_OH=<min grid>];
// tools/wino_stamps.py, tools/ir_stamps.py): stamps per k-block of the workgroup's second item
#ifdef YL_WINO_STAMP
#ifndef YL_STAMP_OH
#define YL_STAMP_OH 80
#endif
__device__ unsigned long long yl_wino_stamps[256 * 8 * 64];
#define WINO_STAMP(i)                                                                                            \
  do {                                                                                                           \
    if (p.Cin == YL_WINO_STAMP && p.OH >= YL_STAMP_OH && lane == 0 && blockIdx.x < 256 && wi == 1 && (i) < 64)     \
      yl_wino_stamps[((size_t)blockIdx.x * 8 + wave) * 64 + (i)] = __builtin_readcyclecounter();                \
  } while (0)
extern "C" int yl_debug_wino_stamps(void* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(yl_wino_stamps), sizeof(yl_wino_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define WINO_STAMP(i) do {} while (0)
#endif

template <int NTW>
struct YlDwcCfg {
  static constexpr int KBMAX = NTW == 1 ? 18 : NTW == 2 ? 9 : NTW == 3 ? 6 : 4;
};

#define YL_SELECT_PROBLEM_C(m)                                                      \
  int yl_k = 0;                                                                     \
  if ((m).n > 1 && (int)blockIdx.x >= (m).p[1].blk0) yl_k = 1;                     \
  if ((m).n > 2 && (int)blockIdx.x >= (m).p[2].blk0) yl_k = 2;                     \
  if ((m).n > 3 && (int)blockIdx.x >= (m).p[3].blk0) yl_k = 3;                     \
  const YlConvP& p = (m).p[yl_k];                                                   \
  const int bx = p.nblk ? (int)blockIdx.x - p.blk0 : (int)blockIdx.x;               \
  const int gx = p.nblk ? p.nblk : (int)gridDim.x;

// patch row pitch in floats (32 channels per pixel): HP pixels + padding chosen so that the 16 lanes of every
// ds_read_b128 service group -- (row py, pixel pair xh, channel half) combinations -- hit 16 distinct 16-byte slots
template <int DK, int DS>
struct YlDwcGeo {
  static constexpr int HP = 3 * DS + DK;
  static constexpr int PITCH = DS == 1 ? HP * 32 + ((HP * 128) % 256 == 128 ? 0 : 32) : HP * 32 + 16;
  static constexpr int HALF_F = HP * PITCH;                    // floats of one half patch (32 channels)
  static constexpr int RI = (HP + 7) / 8;                      // LDS-DMA instructions per patch row (8 pixels x 128 B each)
  static constexpr int NDMA = HP * RI;                         // instructions per half patch
};

template <int DK, int DS, int NTW>
__global__ __launch_bounds__(512, 2) void yl_conv_dwc_kernel(YlConvMulti mp) {
  YL_SELECT_PROBLEM_C(mp)
  using G = YlDwcGeo<DK, DS>;
  constexpr int KBMAX = YlDwcCfg<NTW>::KBMAX;
  constexpr int HP = G::HP, PITCH = G::PITCH, NR = DS + DK;     // NR: patch columns a lane reads per tap row
  extern __shared__ __attribute__((aligned(16))) float yl_clds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < 4;
  const int KB = p.KB;
  const int NG = (KB + 3) >> 2;                                 // 64-channel groups
  // LDS carve: [2][KB][64] float4 B fragments | [DK*DK][Cin] taps, [Cin] bias | one half patch per producer wave
  f32x4* bbuf = reinterpret_cast<f32x4*>(yl_clds);
  float* dwl = yl_clds + (size_t)2 * KB * 256;
  float* halo0 = dwl + (((size_t)(DK * DK + 1) * p.Cin + 3) & ~(size_t)3);
  // every parameter the loops need, read ONCE: `p` points into the kernel-argument segment (problem selected at
  // run time), and the compiler re-reads such fields with s_load + s_waitcnt lgkmcnt(0) wherever it runs short of
  // SGPRs -- inside the tap / copy loops that was most of their time
  const int Cin = p.Cin, H = p.H, W = p.W, OH = p.OH, OW = p.OW, N = p.N, NTtot = p.NTtot, dw_act = p.dw_act;
  const int pad_t = p.dw_pad_t, pad_l = p.dw_pad_l;
  const yl_act_t* const xin = p.x;
  const long zdelta = p.zeros - p.x;                            // float offset of the zero buffer from the input tensor
  const int tw = OW >> 2, th = OH >> 2;
  const int tiles_img = tw * th;
  const int ntiles = p.B * tiles_img;
  // Tiles of this workgroup: t0, t0 + tstride, ... (nmine of them), tile = (image * th + tile row) * tw + tile column.
  // XCD-aware when the grid allows it (gx % 8 == 0 and, in a level-batched launch, the problem starts at a multiple
  // of 8): workgroup b runs on XCD b % 8 (observed dispatch order; a different placement changes speed only), and
  // XCD x owns the contiguous band [x*U/8, (x+1)*U/8) of the U = B * th tile rows -- whole images for B % 8 == 0, the
  // SAME images in every layer, so the halo overlap of neighbouring tiles and the producer layer's output are found
  // in this XCD's L2 instead of crossing the fabric from another XCD's.
  int t0, tstride, nmine;
  if ((gx & 7) == 0 && ((int)blockIdx.x & 7) == (bx & 7)) {
    const int U = p.B * th, x = bx & 7, j = bx >> 3, nj = gx >> 3;
    const int r0 = (int)(((long)U * x) >> 3), r1 = (int)(((long)U * (x + 1)) >> 3);
    const int cnt = (r1 - r0) * tw;
    t0 = r0 * tw + j; tstride = nj;
    nmine = j < cnt ? (cnt - 1 - j) / nj + 1 : 0;
  } else {
    t0 = bx; tstride = gx;
    nmine = bx < ntiles ? (ntiles - 1 - bx) / gx + 1 : 0;
  }

  DWC_STAMP(0);
  {                                                              // depthwise taps + bias -> LDS (asynchronous)
    const int nw = DK * DK * Cin;
    yl_glds_floats(p.dw_w, dwl, nw, tid, 512);
    if (p.dw_b) yl_glds_floats(p.dw_b, dwl + nw, Cin, tid, 512);
    else for (int i = tid; i < Cin; i += 512) dwl[nw + i] = 0.0f;
  }

  if (producer) {
    // =================================================================================== depthwise waves
    // lane = (output row py, pixel pair xh, 4 of the half group's 32 channels): 2 output pixels per lane, NR float4
    // patch reads + DK tap reads per tap row.  Steps of a tile: (group g = wave, wave + 4, ...) x (half 0, 1).
    const int py = lane >> 4, xh = (lane >> 3) & 1, c8 = lane & 7;
    float* hreg = halo0 + wave * G::HALF_F;
    const float dlo = (dw_act == YL_ACT_RELU || dw_act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
    const float dhi = (dw_act == YL_ACT_RELU6) ? 6.0f : INFINITY;
    // half patch (32 channels from c0) of a tile: HP rows x RI float4 per lane (lane = (pixel column lane >> 3 [+ 8q],
    // 4 channels)), fetched into registers one step ahead and written to the wave's LDS patch right before its taps.
    // (Asynchronous global->LDS copies were measured first: ~150-230 issue cycles per 1-KiB piece on the issuing wave,
    // 1200-1900 cycles per half patch -- more than the taps themselves; loads + ds_write_b128 cost a fraction.)
    // Pixels outside the image and channels beyond Cin come from a zero buffer (the depthwise zero padding).
    f32x4 stg[HP * G::RI];
    auto load_half = [&](int tile, int c0) {
      const int b = tile / tiles_img;
      const int trem = tile - b * tiles_img;
      const int tyi = trem / tw, txi = trem - tyi * tw;
      const int iy0 = 4 * tyi * DS - pad_t, ix0 = 4 * txi * DS - pad_l;
      const int cc = c0 + (lane & 7) * 4;
      const bool cok = cc < Cin;
      const long img = (long)b * H * W * Cin + cc;
      const long zoff = zdelta + (lane & 7) * 4;
      long coloff[G::RI];
      bool colok[G::RI];
#pragma unroll
      for (int q = 0; q < G::RI; ++q) {
        const int ix = ix0 + q * 8 + (lane >> 3);
        colok[q] = cok && ix >= 0 && ix < W && q * 8 + (lane >> 3) < HP;
        coloff[q] = img + (long)ix * Cin;
      }
#pragma unroll
      for (int r = 0; r < HP; ++r) {
        const int iy = iy0 + r;                                   // wave-uniform
        const bool rowok = iy >= 0 && iy < H;
        const long rowoff = (long)iy * W * Cin;
#pragma unroll
        for (int q = 0; q < G::RI; ++q) {
          const long off = (rowok && colok[q]) ? rowoff + coloff[q] : zoff;     // select, no branch
          stg[r * G::RI + q] = yl_ld4(xin + off);
        }
      }
    };
    auto store_half = [&](float* dst) {
#pragma unroll
      for (int r = 0; r < HP; ++r)
#pragma unroll
        for (int q = 0; q < G::RI; ++q)
          if (q * 8 + 7 < HP || q * 8 + (lane >> 3) < HP)
            *reinterpret_cast<f32x4*>(dst + r * PITCH + q * 256 + lane * 4) = stg[r * G::RI + q];
    };
    // step stream of this wave: s = 0, 1, ... over (tile index, group, half)
    const int gcount = NG > wave ? (NG - 1 - wave) / 4 + 1 : 0;   // groups of this wave per tile
    auto halves_of = [&](int g) { return (4 * g + 2 < KB) ? 2 : 1; };
    int st_it = 0, st_gi = 0, st_h = 0;                           // the step whose patch is requested NEXT
    auto step_valid = [&]() { return gcount > 0 && st_it < nmine; };
    auto step_c0 = [&]() { return (wave + 4 * st_gi) * 64 + st_h * 32; };
    auto step_advance = [&]() {
      if (++st_h >= halves_of(wave + 4 * st_gi)) { st_h = 0; if (++st_gi >= gcount) { st_gi = 0; ++st_it; } }
    };
    if (step_valid()) { load_half(t0 + st_it * tstride, step_c0()); step_advance(); }
    __syncthreads();                                              // taps are in LDS (drains the copy queue once)
    DWC_STAMP(1);
    for (int it = 0; it <= nmine; ++it) {
      if (it < nmine) {
        f32x4* bb = bbuf + (size_t)(it & 1) * KB * 64;
        for (int gi = 0; gi < gcount; ++gi) {
          const int g = wave + 4 * gi;
          const int nh = halves_of(g);
          for (int h = 0; h < nh; ++h) {
            if (it == 2 && gi == 0) DWC_STAMP(20 + 5 * h);
            store_half(hreg);                                       // this step's patch: registers -> LDS (after the
            if (it == 2 && gi == 0) DWC_STAMP(21 + 5 * h);          // previous step's tap reads, in LDS order)
            if (step_valid()) { load_half(t0 + st_it * tstride, step_c0()); step_advance(); }   // next step's patch: in flight under the taps
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // patch writes (all lanes) -> tap reads
            if (it == 2 && gi == 0) DWC_STAMP(22 + 5 * h);
            const float* hp_ = hreg;
            const int c = g * 64 + h * 32 + c8 * 4;
            const int cs = c < Cin ? c : Cin - 4;
            const float* tapw = dwl + cs;                            // tap t of this lane's 4 channels: tapw[t * Cin]
            f32x4 o[2];
            o[0] = o[1] = yl_ld4(tapw + DK * DK * Cin);
            const float* hrow = hp_ + (size_t)(py * DS) * PITCH + (2 * xh * DS) * 32 + c8 * 4;
            // tap rows: row dy+1 is read while row dy is multiplied (two rows of registers; unrolling all DK rows
            // spills for 5x5)
            f32x4 vn[NR], wn[DK];
#pragma unroll
            for (int x = 0; x < NR; ++x) vn[x] = *reinterpret_cast<const f32x4*>(hrow + x * 32);
#pragma unroll
            for (int dx = 0; dx < DK; ++dx) wn[dx] = yl_ld4(tapw + dx * Cin);
#pragma unroll 1
            for (int dy = 0; dy < DK; ++dy) {
              f32x4 v[NR], w[DK];
#pragma unroll
              for (int x = 0; x < NR; ++x) v[x] = vn[x];
#pragma unroll
              for (int dx = 0; dx < DK; ++dx) w[dx] = wn[dx];
              if (dy + 1 < DK) {
#pragma unroll
                for (int x = 0; x < NR; ++x) vn[x] = *reinterpret_cast<const f32x4*>(hrow + (dy + 1) * PITCH + x * 32);
#pragma unroll
                for (int dx = 0; dx < DK; ++dx) wn[dx] = yl_ld4(tapw + ((dy + 1) * DK + dx) * Cin);
              }
#pragma unroll
              for (int dx = 0; dx < DK; ++dx)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                  const f32x4 a = v[j * DS + dx];
                  o[j].x = fmaf(a.x, w[dx].x, o[j].x); o[j].y = fmaf(a.y, w[dx].y, o[j].y);
                  o[j].z = fmaf(a.z, w[dx].z, o[j].z); o[j].w = fmaf(a.w, w[dx].w, o[j].w);
                }
            }
            // B fragment of channel block kb: lane (kq = c8 & 3, pixel py*4 + 2*xh + j).  Blocks beyond KB (group
            // tail) are not stored; channels beyond Cin inside a block meet zero 1x1 weights.
            if (it == 2 && gi == 0) { asm volatile("" :: "v"(o[0].x), "v"(o[1].x)); DWC_STAMP(23 + 5 * h); }
            const int kb = 4 * g + 2 * h + (c8 >> 2);
            if (kb < KB) {
#pragma unroll
              for (int j = 0; j < 2; ++j)
                bb[kb * 64 + (c8 & 3) * 16 + py * 4 + 2 * xh + j] = yl_actc(o[j], dw_act, dlo, dhi);
            }
            if (it == 2 && gi == 0) DWC_STAMP(24 + 5 * h);
          }
        }
      }
      DWC_STAMP(2 + 2 * it);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // B fragments written; the patch copy stays in flight
      __builtin_amdgcn_s_barrier();
      DWC_STAMP(3 + 2 * it);
    }
  } else {
    // =================================================================================== GEMM waves
    const int cw = wave - 4;
    const int kq = lane >> 4, pl = lane & 15;
    const int nt0 = cw * NTW;
    const bool p2 = nt0 < NTtot;                                   // this wave owns output channels
    const yl_act_t* const resp = p.res;
    yl_act_t* const outp = p.out;
    const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
    const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
    f32x4 wreg[KBMAX][NTW], breg[NTW];
    {
      const f32x4* wg = reinterpret_cast<const f32x4*>(p.wp);
#pragma unroll
      for (int kb = 0; kb < KBMAX; ++kb)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const bool ok = kb < KB && nt0 + nt < NTtot;
          wreg[kb][nt] = ok ? wg[((size_t)kb * NTtot + nt0 + nt) * 64 + lane] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) breg[nt] = yl_ld4(p.bias + (nt0 + nt) * 16 + 4 * kq);   // bias is padded
    }
    const bool pre_add = p.res != nullptr && p.up == nullptr && p.act == YL_ACT_NONE;
    const bool generic = !pre_add && (p.res || p.up || YL_SMOOTH(p.act));
    __syncthreads();
    DWC_STAMP(1);
    for (int it = 0; it <= nmine; ++it) {
      if (it >= 1 && p2) {
        const int tile = t0 + (it - 1) * tstride;
        const f32x4* bb = bbuf + (size_t)((it - 1) & 1) * KB * 64;
        const int b = tile / tiles_img;
        const int trem = tile - b * tiles_img;
        const int tyi = trem / tw, txi = trem - tyi * tw;
        YlPix px[1];
        px[0].b = b;
        px[0].oy = 4 * tyi + (pl >> 2);
        px[0].ox = 4 * txi + (pl & 3);
        px[0].valid = true;
        px[0].lin = ((size_t)b * OH + px[0].oy) * OW + px[0].ox;
        f32x4 acc[1][NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int n = (nt0 + nt) * 16 + 4 * kq;
          acc[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (pre_add && n < N) acc[0][nt] = yl_ld4(resp + px[0].lin * N + n);
        }
        if (it == 3) { asm volatile("" :: "v"(acc[0][0].x)); DWC_STAMP(20); }
        // B fragments are read PF channel blocks ahead of their MFMAs (one ds_read_b128 each; the depthwise waves
        // keep the LDS pipe busy, so a read issued one block ahead returned too late)
        constexpr int PF = KBMAX < 4 ? KBMAX : 4;
        f32x4 xr[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) xr[i] = bb[(i < KB ? i : 0) * 64 + lane];
#pragma unroll
        for (int kb = 0; kb < KBMAX; ++kb) {
          if (kb < KB) {
            f32x4 xq[1];
            xq[0] = xr[kb % PF];
            if (kb + PF < KBMAX) xr[kb % PF] = bb[(kb + PF < KB ? kb + PF : 0) * 64 + lane];
            yl_mma_step<NTW, 1>(wreg[kb], xq, acc);
          }
        }
        if (it == 3) { asm volatile("" :: "v"(acc[0][0].x)); DWC_STAMP(21); }
        if (generic) yl_epi_generic<NTW, 1>(p, acc, px, nt0, kq);
        else {                                                      // == yl_epi_fast with the bias from registers
          yl_act_t* orow = outp + px[0].lin * N;
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            const int n = (nt0 + nt) * 16 + 4 * kq;
            const f32x4 v = yl_clamp4(acc[0][nt] + breg[nt], lo, hi);
            if (n < N) yl_st4(orow + n, v);
          }
        }
      }
      DWC_STAMP(2 + 2 * it);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this tile's B-fragment reads are complete
      __builtin_amdgcn_s_barrier();
      DWC_STAMP(3 + 2 * it);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Wave-autonomous 1x1 convolution for launches whose pixel count cannot fill the chip with 4-wave tiles
// (40x40 / 20x20 / 10x10 grids).  yl_conv_mfma_kernel (yl_conv.hip) is built for streaming many tiles per
// workgroup: persistent grid, weight image copied to LDS first, a wave walks all n-tiles of its pixels.  At
// M = 25 600 pixels that is 400 workgroups that spend their life in the weight prologue and one dependent tile:
// 25-47 us for layers whose MFMA time is 2-5 us.  Here every WAVE is an independent work item -- 16*MT pixels x
// NTW n-tiles -- with no LDS and no barrier: the A fragments (weights) and B fragments (activations) come straight
// from global memory (L1/L2-resident: the 4 waves of a workgroup share one pixel group, all workgroups share the
// weights), ~70 VGPRs, so 6-8 waves per SIMD cover each other's load latency and MFMA dependency chains, and the
// launch has thousands of waves instead of a few hundred.  Same k order, same epilogues as yl_conv_mfma_kernel:
// bit-identical results.
// SC: the input is multiplied by a squeeze-excite gate [B][Cin] (YlConvP::scale) in the B-operand path -- x * gate in one
// fp32 rounding, then the GEMM (timm: `x * gate` feeds conv_pwl).
template <int NTW, int MT, bool DEC, bool SC = false>
__global__ __launch_bounds__(256) void yl_conv_pwt_kernel(YlConvMulti mp, int nchunk) {
  YL_SELECT_PROBLEM_C(mp)                                   // level-batched launches: block ranges per problem
  (void)gx;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  const int item = bx * 4 + wave;
  const int mg = item / nchunk, nc = item - mg * nchunk;
  const int nt0 = nc * NTW;
  const int Cin = p.Cin, N = p.N, NTtot = p.NTtot, KB = p.KB, M = p.M;
  if ((long)mg * (MT * 16) >= M) return;
  const yl_act_t* const xin = p.x;
  const f32x4* const wg = reinterpret_cast<const f32x4*>(p.wp);
  YlPix px[MT];
  const yl_act_t* xrow[MT];
  const float* srow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    size_t lin = (size_t)mg * (MT * 16) + mt * 16 + pl;
    px[mt].valid = lin < (size_t)M;
    if (!px[mt].valid) lin = (size_t)M - 1;
    px[mt].lin = lin;
    px[mt].b = 0; px[mt].oy = 0; px[mt].ox = 0;
    if (DEC || SC || p.up) {                                // only the upsample-add / decode epilogues / the gate need coordinates
      const int ohw = p.OH * p.OW;
      const int b = (int)(lin / ohw);
      const int rem = (int)(lin - (size_t)b * ohw);
      px[mt].b = b;
      px[mt].oy = rem / p.OW;
      px[mt].ox = rem - px[mt].oy * p.OW;
    }
    xrow[mt] = xin + lin * Cin + 4 * kq;
    srow[mt] = SC ? p.scale + (size_t)px[mt].b * Cin + 4 * kq : nullptr;
  }
  // n-tiles beyond the layer's last one (partial last chunk): read the last tile's weights, never stored
  int wofs[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) wofs[nt] = ((nt0 + nt < NTtot ? nt0 + nt : NTtot - 1) * 64 + lane);
  f32x4 acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // residual / upsample-add without activation: the addends initialise the accumulators (as yl_conv_mfma_kernel)
  const bool pre_add = (p.res || p.up) && p.act == YL_ACT_NONE;
  if (pre_add) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const size_t obase = px[mt].lin * N;
      size_t up_off = 0;
      if (p.up) {
        const int uy = (px[mt].oy * p.UH) / p.OH, ux = (px[mt].ox * p.UW) / p.OW;
        up_off = (((size_t)px[mt].b * p.UH + uy) * p.UW + ux) * N;
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int n = (nt0 + nt) * 16 + 4 * kq;
        if (n < N) {
          if (p.res) acc[mt][nt] = yl_ld4(p.res + obase + n);
          if (p.up) acc[mt][nt] += yl_ld4(p.up + up_off + n);
        }
      }
    }
  }
  constexpr int UK = 2;
  const int cin4 = Cin - 4;
  for (int kb0 = 0; kb0 < KB; kb0 += UK) {
    f32x4 a[UK][NTW], bq[UK][MT];
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      const int kb = kb0 + u < KB ? kb0 + u : KB - 1;       // odd KB: the tail step re-reads the last block, unused
      const int c = kb * 16 + 4 * kq;
      const bool cok = c < Cin;                              // channel tail of the last block: zero activations
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) a[u][nt] = wg[(size_t)kb * NTtot * 64 + wofs[nt]];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        bq[u][mt] = yl_ld4(cok ? xrow[mt] + kb * 16 : p.zeros);
        if (SC) bq[u][mt] *= yl_ld4(cok ? srow[mt] + kb * 16 : reinterpret_cast<const float*>(p.zeros));
      }
    }
    (void)cin4;
#pragma unroll
    for (int u = 0; u < UK; ++u)
      if (kb0 + u < KB) yl_mma_step<NTW, MT>(a[u], bq[u], acc);
  }
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  if (DEC) { yl_epi_decode<NTW, MT, false, true>(p, acc, px, 0, kq, lane); return; }   // (nchunk == 1: nt0 == 0)
    // head output under yl_predict (one wave = whole rows)
  if (!pre_add && (p.res || p.up || YL_SMOOTH(p.act))) yl_epi_generic<NTW, MT>(p, acc, px, nt0, kq);
  else yl_epi_fast<NTW, MT>(p, acc, px, nt0, kq, lo, hi, true);
}

template <int NTW, int MT, bool DEC = false, bool SC = false>
static hipError_t pwt_go(YlConvMulti& m, int nchunk, hipStream_t st) {
  int at = 0;
  for (int k = 0; k < m.n; ++k) {
    const long groups = ((long)m.p[k].M + MT * 16 - 1) / (MT * 16);
    const long items = groups * nchunk;
    m.p[k].blk0 = at;
    m.p[k].nblk = (int)((items + 3) / 4);
    at += m.p[k].nblk;
  }
  if (m.n == 1) m.p[0].nblk = 0;                                 // single problem: the whole grid (YL_SELECT_PROBLEM_C)
  hipLaunchKernelGGL((yl_conv_pwt_kernel<NTW, MT, DEC, SC>), dim3((unsigned)at), dim3(256), 0, st, m, nchunk);
  return hipGetLastError();
}

// n <= 4 plain 1x1 stride-1 convs of identical configuration (one launch).  Either N % 4 == 0 (float4 epilogues),
// or head-output layers whose decode runs in the epilogue and whose raw rows are not wanted (yl_predict, no mask
// coefficients): then one wave holds whole rows (N <= 96).  hipErrorNotSupported otherwise.
hipError_t yl_launch_conv_pwt_multi(const YlConvP* ps, int n, hipStream_t st) {
  if (n < 1 || n > 4) return hipErrorNotSupported;
  YlConvMulti m = {};
  m.n = n;
  long Mtot = 0;
  for (int k = 0; k < n; ++k) {
    const YlConvP& p = ps[k];
    if (p.k != 1 || p.stride != 1 || p.dw_k > 0 || p.C1 > 0 || p.in_shift) return hipErrorNotSupported;
    if (p.scale && (n != 1 || p.dec_boxes)) return hipErrorNotSupported;      // gated input: single plain layer
    const bool dec = p.dec_boxes && !p.dec_raw;
    if ((p.N & 3) && !dec) return hipErrorNotSupported;
    if (p.dec_boxes && !dec) return hipErrorNotSupported;
    if (dec && p.NTtot > 6) return hipErrorNotSupported;
    if (p.NTtot != ps[0].NTtot || p.KB != ps[0].KB || (p.dec_boxes != nullptr) != (ps[0].dec_boxes != nullptr)) return hipErrorNotSupported;
    m.p[k] = p;
    Mtot += p.M;
  }
  const YlConvP& p = ps[0];
  const int NT = p.NTtot;
  const bool dec = p.dec_boxes != nullptr;
  // n-tiles per wave: as many as 4 (the activations are fetched once per wave), chunks of equal size; decode: the row
  int ntw = dec ? (NT <= 4 ? NT : 6) : (NT <= 4 ? NT : (NT % 4 == 0 ? 4 : (NT % 3 == 0 ? 3 : 4)));
  const int nchunk = dec ? 1 : (NT + ntw - 1) / ntw;
  const bool two = Mtot >= 65536;                               // two m-tiles per wave halve the weight traffic
  if (dec) {
    switch (ntw) {
      case 1: return pwt_go<1, 1, true>(m, nchunk, st);
      case 2: return pwt_go<2, 1, true>(m, nchunk, st);
      case 3: return pwt_go<3, 1, true>(m, nchunk, st);
      case 4: return pwt_go<4, 1, true>(m, nchunk, st);
      default: return two ? pwt_go<6, 2, true>(m, nchunk, st) : pwt_go<6, 1, true>(m, nchunk, st);
    }
  }
  if (p.scale) {
    switch (ntw) {
      case 1: return pwt_go<1, 1, false, true>(m, nchunk, st);
      case 2: return pwt_go<2, 1, false, true>(m, nchunk, st);
      case 3: return pwt_go<3, 1, false, true>(m, nchunk, st);
      default: return pwt_go<4, 1, false, true>(m, nchunk, st);
    }
  }
  switch (ntw) {
    case 1: return two ? pwt_go<1, 2>(m, nchunk, st) : pwt_go<1, 1>(m, nchunk, st);
    case 2: return two ? pwt_go<2, 2>(m, nchunk, st) : pwt_go<2, 1>(m, nchunk, st);
    case 3: return two ? pwt_go<3, 2>(m, nchunk, st) : pwt_go<3, 1>(m, nchunk, st);
    default: return two ? pwt_go<4, 2>(m, nchunk, st) : pwt_go<4, 1>(m, nchunk, st);
  }
}

hipError_t yl_launch_conv_pwt(const YlConvP& p, hipStream_t st) { return yl_launch_conv_pwt_multi(&p, 1, st); }

// ------------------------------------------------------------------------------------------------
// Wave-autonomous depthwise -> 1x1 convolution: the depthwise counterpart of yl_conv_pwt_kernel and the successor of
// yl_conv_dwh_kernel (same per-wave algorithm: 4x4-pixel m-tiles, per 16-channel block the halo patch goes through
// the wave's private LDS region, B = act(bias + sum_taps w*x), NT x 4 MFMAs per m-tile; same k order and epilogues:
// bit-identical).  Persistent workgroups of four waves, one barrier (prologue), tiles dealt in XCD bands.  Two knobs:
//   WL   where the 1x1 A fragments come from.  false: straight from L1/L2 (requested before the taps of the block
//        they belong to) -- ~15 KB of LDS per workgroup, so K >= 192 layers keep 3-4 waves per SIMD where the LDS
//        weight image allowed one workgroup per CU (50 -> 36 us);  true: an LDS image filled once per workgroup,
//        the better choice when it is small (K = N = 96: 36 KB).
//   MT   m-tiles per wave: 2 = a 4x8-pixel tile, both m-tiles share the 6x10 / 8x12 halo patch, every A fragment
//        and every tap-weight read (LDS reads per pixel -35 %).
//   SK   split-K for grids of <= 20 x 20 pixels (round 4): such a layer has 25 tiles per image -- 800 wave tiles for a
//        32-image chunk on 1024 SIMDs, one wave walking 12-18 k-blocks serially while most of the chip idles (the 20x20
//        stage of edge_n: 4x off its byte / MFMA floor, and its launches cost their full duration in the two-stream step:
//        profiles/r04_skip_layers_edge_n_b64.txt).  SK = 4: the four waves of a workgroup share ONE tile, wave w takes the
//        k-blocks w, w + 4, ... (its own halo staging, no barrier in the loop); the partial accumulators meet in LDS (one
//        barrier per tile, buffers alternate) and wave w finishes n-tile w (+4): partials added in wave order 0..3, then
//        the usual epilogue.  Chosen by the LAYER SHAPE only (never by the batch): results stay batch-invariant and
//        bitwise repeatable, but are another fp32 summation order of the same products than SK = 1 / yl_conv_dwh_kernel.
template <int NT, int DK, int DS, int MT, bool WL, int SK = 1>
__global__ __launch_bounds__(256, MT == 2 ? 2 : 3) void yl_conv_dwt_kernel(YlConvMulti mp) {
  YL_SELECT_PROBLEM_C(mp)
  constexpr int HPY = 3 * DS + DK, HPX = (4 * MT - 1) * DS + DK;     // halo patch rows / columns
  constexpr int PITCHF = ((HPX * 16 + 7) / 64) * 64 + 56;            // row pitch in floats (see yl_conv_dwh_kernel)
  constexpr int HF4 = HPY * HPX * 4;
  constexpr int NSLOT = (HF4 + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) float yl_clds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  const int Cin = p.Cin, H = p.H, W = p.W, OH = p.OH, OW = p.OW, N = p.N, NTtot = p.NTtot, KB = p.KB;
  const yl_act_t* const xin = p.x;
  // staging loads through a raw buffer descriptor (round 6, see yl_conv_dws_kernel): 32-bit byte offset per slot + scalar k-block
  // offset, zeros outside the image from the hardware range check; the channel tail is not masked (the 1x1 weights of those k
  // slots are zeros, the tap weights are clamped to the last channels, the arenas end in 256 spare bytes)
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<yl_act_t*>(p.x), 0, (int)((long)p.B * H * W * Cin * (long)sizeof(yl_act_t)), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  f32x4* wl = reinterpret_cast<f32x4*>(yl_clds);                     // WL: [KB][NTtot][64] float4
  float* dwl = yl_clds + (WL ? (size_t)KB * NTtot * 256 : 0);        // [DK*DK][Cin] taps, [Cin] bias
  float* halo = dwl + (((size_t)(DK * DK + 1) * Cin + 3) & ~(size_t)3) + wave * (HPY * PITCHF);
  f32x4* red = reinterpret_cast<f32x4*>(dwl + (((size_t)(DK * DK + 1) * Cin + 3) & ~(size_t)3) + 4 * (HPY * PITCHF));   // SK: [2][4][NT][64]
  const f32x4* wg = reinterpret_cast<const f32x4*>(p.wp);
  const int twn = OW / (4 * MT), thn = OH >> 2;
  const int tiles_img = twn * thn;
  const int ntiles = p.B * tiles_img;
  // tile order: workgroup b runs on XCD b % 8; each XCD takes one contiguous band of tiles (halo rows of neighbouring
  // tiles then meet in the same L2), its workgroups' waves interleave inside the band
  int tile, tend, wstride;
  constexpr int WPT = SK > 1 ? 1 : 4;                                // tiles a workgroup works on at a time
  if ((gx & 7) == 0) {
    const int tpx = (ntiles + 7) >> 3;
    const int band0 = (bx & 7) * tpx;
    tend = (band0 + tpx) < ntiles ? (band0 + tpx) : ntiles;
    tile = band0 + (bx >> 3) * WPT + (SK > 1 ? 0 : wave);
    wstride = (gx >> 3) * WPT;
  } else {
    tile = bx * WPT + (SK > 1 ? 0 : wave); tend = ntiles; wstride = gx * WPT;
  }
  const int kb0 = SK > 1 ? wave : 0;                                 // SK: this wave's k-blocks kb0, kb0 + SK, ...
  unsigned sk_par = 0;
  // staging slots of this lane: halo pixel / channel quad -> LDS offset; global offsets per tile
  int s_lo[NSLOT];
  bool s_ok[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int e = j * 64 + lane;
    s_ok[j] = e < HF4;
    const int hp = (s_ok[j] ? e : 0) >> 2, quad = e & 3;
    const int hr = hp / HPX, hc = hp - hr * HPX;
    s_lo[j] = hr * PITCHF + hc * 16 + quad * 4;
  }
  // (image, tile row, tile column) of the wave's current tile, advanced by the decomposition of wstride with two
  // carries: no integer division per tile (each one is ~25 VALU instructions, and fp32 VALU time is MFMA time)
  int tb = 0, tyi = 0, txi = 0;
  if (tile < tend) {
    tb = tile / tiles_img;
    const int trem = tile - tb * tiles_img;
    tyi = trem / twn; txi = trem - tyi * twn;
  }
  const int sdb = wstride / tiles_img, sdrem = wstride - sdb * tiles_img;
  const int sdy = sdrem / twn, sdx = sdrem - sdy * twn;
  unsigned goff[NSLOT];
  auto tile_geom = [&]() {
    const int b = tb;
    const int iy0 = 4 * tyi * DS - p.dw_pad_t, ix0 = 4 * MT * txi * DS - p.dw_pad_l;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int e = j * 64 + lane;
      const int hp = (e < HF4 ? e : 0) >> 2;
      const int hr = hp / HPX, hc = hp - hr * HPX;
      const int iy = iy0 + hr, ix = ix0 + hc;
      const bool in = e < HF4 && iy >= 0 && iy < H && ix >= 0 && ix < W;
      goff[j] = in ? (unsigned)((((b * H + iy) * W + ix) * Cin + (lane & 3) * 4) * (int)sizeof(yl_act_t)) : OOB;
    }
  };
  auto stage_load = [&](int kb, f32x4 (&r)[NSLOT]) {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
#if defined_YL_F16S
      r[j] = __builtin_convertvector(__builtin_bit_cast(yl_h16x4, __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)goff[j], kb * 32, 0)), f32x4);
#else
      r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)goff[j], kb * 64, 0));
#endif
    }
  };
  auto stage_store = [&](const f32x4 (&r)[NSLOT]) {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
      if (s_ok[j]) *reinterpret_cast<f32x4*>(halo + s_lo[j]) = r[j];
  };
  f32x4 stg[NSLOT];
  bool primed = false;
  if (tile < tend) {                       // first tile's first halo block: in flight together with the LDS fills
    tile_geom();
    stage_load(kb0, stg);
    primed = true;
  }
  if (WL) {
    for (int i = wave; i < KB * NTtot; i += 4) yl_glds16(wg + (size_t)i * 64 + lane, wl + (size_t)i * 64);
  }
  {
    const int nw = DK * DK * Cin;
    yl_glds_floats(p.dw_w, dwl, nw, tid, 256);
    if (p.dw_b) yl_glds_floats(p.dw_b, dwl + nw, Cin, tid, 256);
    else for (int i = tid; i < Cin; i += 256) dwl[nw + i] = 0.0f;
  }
  __syncthreads();                                                   // the only workgroup barrier
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float dlo = (p.dw_act == YL_ACT_RELU || p.dw_act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float dhi = (p.dw_act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const int dw_act = p.dw_act;
  const bool pre_add = p.res != nullptr && p.up == nullptr && p.act == YL_ACT_NONE;
  const int rbase = ((pl >> 2) * DS) * PITCHF + ((pl & 3) * DS) * 16 + 4 * kq;
  for (; tile < tend; tile += wstride) {
    const int b = tb;
    YlPix px[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      px[mt].b = b;
      px[mt].oy = 4 * tyi + (pl >> 2);
      px[mt].ox = 4 * MT * txi + 4 * mt + (pl & 3);
      px[mt].valid = true;
      px[mt].lin = ((size_t)b * OH + px[mt].oy) * OW + px[mt].ox;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + 4 * kq;
        acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (pre_add && n < N && (SK == 1 || wave == 0)) acc[mt][nt] = yl_ld4(p.res + px[mt].lin * N + n);
      }
    if (!primed) {
      tile_geom();
      stage_load(kb0, stg);
    }
    primed = false;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // previous tile's tap reads are complete
    stage_store(stg);
    // (Requesting the NEXT tile's first halo block under the last block of this one -- the pipelining that pays in
    // yl_conv_dpp_kernel -- was measured here: 37.54k vs 37.79k images/s on the headline step, slower.  So was chaining
    // the next block's plain 1x1 `pw_exp` behind this kernel's epilogue with both outputs written, for the five
    // `pw_proj` -> `pw_exp` pairs of edge_n's 20x20 stage: one-stream launch times at B = 64, 45.4 us chained against
    // 29.4 + 20.2 for the 5x5 pairs, 53.6 against 23.7 + 20.2 for the 3x3 pairs (one of them feeds 64 -> 480); headline
    // unchanged at 38.8k with five launches fewer per chunk -- not kept.)
    for (int kb = kb0; kb < KB; kb += SK) {
      const bool more = kb + SK < KB;
      if (more) stage_load(kb + SK, stg);
      f32x4 wq[NT];
      if (!WL) {                                                     // A fragments from L1/L2: in flight under the taps
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wq[nt] = wg[((size_t)kb * NTtot + (nt < NTtot ? nt : NTtot - 1)) * 64 + lane];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");        // halo writes (all lanes) -> tap reads
      const int c = kb * 16 + 4 * kq;
      const int cs = c < Cin ? c : Cin - 4;
      const float* tapw = dwl + cs;
      f32x4 xq[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xq[mt] = yl_ld4(tapw + DK * DK * Cin);
      if (DK == 3) {
#pragma unroll
        for (int dy = 0; dy < DK; ++dy) {
#pragma unroll
          for (int dx = 0; dx < DK; ++dx) {
            const f32x4 w = yl_ld4(tapw + (dy * DK + dx) * Cin);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(halo + rbase + dy * PITCHF + (dx + 4 * mt * DS) * 16);
              xq[mt].x = fmaf(v.x, w.x, xq[mt].x); xq[mt].y = fmaf(v.y, w.y, xq[mt].y);
              xq[mt].z = fmaf(v.z, w.z, xq[mt].z); xq[mt].w = fmaf(v.w, w.w, xq[mt].w);
            }
          }
          if (MT > 1) __builtin_amdgcn_sched_barrier(0);             // one tap row in flight: bounds the live LDS reads
        }
      } else {
#pragma unroll 1
        for (int dy = 0; dy < DK; ++dy) {                            // one tap row at a time bounds the register footprint
#pragma unroll
          for (int dx = 0; dx < DK; ++dx) {
            const f32x4 w = yl_ld4(tapw + (dy * DK + dx) * Cin);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(halo + rbase + dy * PITCHF + (dx + 4 * mt * DS) * 16);
              xq[mt].x = fmaf(v.x, w.x, xq[mt].x); xq[mt].y = fmaf(v.y, w.y, xq[mt].y);
              xq[mt].z = fmaf(v.z, w.z, xq[mt].z); xq[mt].w = fmaf(v.w, w.w, xq[mt].w);
            }
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xq[mt] = yl_actc(xq[mt], dw_act, dlo, dhi);
      // channel tail (c >= Cin): the packed 1x1 weights of those k slots are zero, no select needed
      if (WL) {
        const f32x4* wrow = wl + (size_t)kb * NTtot * 64 + lane;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wq[nt] = wrow[(nt < NTtot ? nt : NTtot - 1) * 64];
      }
      yl_mma_step<NT, MT>(wq, xq, acc);
      if (more) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // this block's tap reads are complete
        stage_store(stg);
      }
    }
    if (SK > 1) {
      // partial accumulators -> LDS (buffer = tile parity), ONE barrier per tile, wave w finishes the n-tiles w, w + SK
      f32x4* rb = red + (size_t)(sk_par & 1u) * (4 * NT * 64);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) rb[(wave * NT + nt) * 64 + lane] = acc[0][nt];
      __syncthreads();
      ++sk_par;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if ((nt & (SK - 1)) != wave) continue;                       // wave-uniform
        f32x4 one[1][1];
        one[0][0] = ((rb[(0 * NT + nt) * 64 + lane] + rb[(1 * NT + nt) * 64 + lane]) + rb[(2 * NT + nt) * 64 + lane]) +
                    rb[(3 * NT + nt) * 64 + lane];
        const YlPix px1[1] = {px[0]};
        if (!pre_add && (p.res || p.up || YL_SMOOTH(p.act))) yl_epi_generic<1, 1>(p, one, px1, nt, kq);
        else yl_epi_fast<1, 1>(p, one, px1, nt, kq, lo, hi, true);
      }
    } else {
      if (!pre_add && (p.res || p.up || YL_SMOOTH(p.act))) yl_epi_generic<NT, MT>(p, acc, px, 0, kq);
      else yl_epi_fast<NT, MT>(p, acc, px, 0, kq, lo, hi, true);
    }
    txi += sdx;
    if (txi >= twn) { txi -= twn; ++tyi; }
    tyi += sdy;
    if (tyi >= thn) { tyi -= thn; ++tb; }
    tb += sdb;
  }
}

template <int NT, int DK, int DS, int MT, bool WL, int SK = 1>
static hipError_t dwt_go(YlConvMulti& m, hipStream_t st, bool attr_only) {
  if (attr_only)
    return hipFuncSetAttribute((const void*)yl_conv_dwt_kernel<NT, DK, DS, MT, WL, SK>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  constexpr int HPY = 3 * DS + DK, HPX = (4 * MT - 1) * DS + DK;
  constexpr int PITCHF = ((HPX * 16 + 7) / 64) * 64 + 56;
  const YlConvP& p = m.p[0];
  const size_t lds = ((WL ? (size_t)p.KB * p.NTtot * 256 : 0) + (((size_t)(DK * DK + 1) * p.Cin + 3) & ~(size_t)3) +
                      (size_t)4 * HPY * PITCHF + (SK > 1 ? (size_t)2 * 4 * NT * 256 : 0)) * 4;
  if (lds > 96 * 1024) return hipErrorNotSupported;
  const int res = yl_resident_blocks_n(yl_conv_dwt_kernel<NT, DK, DS, MT, WL, SK>, 256, lds);
  long tiles[4], total = 0;
  for (int k = 0; k < m.n; ++k) {
    tiles[k] = (long)m.p[k].B * (m.p[k].OH >> 2) * (m.p[k].OW / (4 * MT)) * (SK > 1 ? 4 : 1);     // SK: a workgroup per tile
    total += tiles[k];
  }
  // persistent workgroups: the co-resident count shared between the problems in proportion to their tiles, each
  // share a multiple of 8 (XCD bands) and at most one wave per tile
  int at = 0;
  for (int k = 0; k < m.n; ++k) {
    long g = ((long)res * tiles[k] / total + 7) & ~7L;
    const long cap = ((tiles[k] + 3) / 4 + 7) & ~7L;
    if (g > cap) g = cap;
    if (g < 8) g = 8;
    m.p[k].blk0 = at;
    m.p[k].nblk = (int)g;
    at += (int)g;
  }
  if (m.n == 1) m.p[0].nblk = 0;
  hipLaunchKernelGGL((yl_conv_dwt_kernel<NT, DK, DS, MT, WL, SK>), dim3((unsigned)at), dim3(256), lds, st, m);
  return hipGetLastError();
}

// split-K form (see the kernel): single problem, N = 49..64 (4 n-tiles, one per wave), stride-1 3x3 / 5x5 or stride-2
// 3x3 depthwise, weights from L1/L2, >= 8 k-blocks, a grid of at most 20 x 20 output pixels
static hipError_t dwt_splitk(YlConvMulti& m, hipStream_t st, bool attr_only) {
  if (attr_only) {
    hipError_t e = dwt_go<4, 3, 1, 1, false, 4>(m, st, true);
    if (e == hipSuccess) e = dwt_go<4, 5, 1, 1, false, 4>(m, st, true);
    if (e == hipSuccess) e = dwt_go<4, 3, 2, 1, false, 4>(m, st, true);
    return e;
  }
  const YlConvP& p = m.p[0];
  if (p.dw_k == 3 && p.dw_stride == 1) return dwt_go<4, 3, 1, 1, false, 4>(m, st, false);
  if (p.dw_k == 5 && p.dw_stride == 1) return dwt_go<4, 5, 1, 1, false, 4>(m, st, false);
  if (p.dw_k == 3 && p.dw_stride == 2) return dwt_go<4, 3, 2, 1, false, 4>(m, st, false);
  return hipErrorNotSupported;
}

template <int NT, int MT, bool WL>
static hipError_t dwt_dk(YlConvMulti& m, hipStream_t st, bool attr_only) {
  const YlConvP& p = m.p[0];
  if (attr_only) {
    hipError_t e = dwt_go<NT, 3, 1, MT, WL>(m, st, true);
    if (e == hipSuccess) e = dwt_go<NT, 3, 2, MT, WL>(m, st, true);
    if (e == hipSuccess) e = dwt_go<NT, 5, 1, MT, WL>(m, st, true);
    if (e == hipSuccess) e = dwt_go<NT, 5, 2, MT, WL>(m, st, true);
    return e;
  }
  if (p.dw_k == 3 && p.dw_stride == 1) return dwt_go<NT, 3, 1, MT, WL>(m, st, false);
  if (p.dw_k == 3 && p.dw_stride == 2) return dwt_go<NT, 3, 2, MT, WL>(m, st, false);
  if (p.dw_k == 5 && p.dw_stride == 1) return dwt_go<NT, 5, 1, MT, WL>(m, st, false);
  if (p.dw_k == 5 && p.dw_stride == 2) return dwt_go<NT, 5, 2, MT, WL>(m, st, false);
  return hipErrorNotSupported;
}

template <int NT>
static hipError_t dwt_nt(YlConvMulti& m, hipStream_t st, bool two, bool wl, bool attr_only) {
#ifdef YL_DWT_MT2          // 4x8-pixel wave tiles: measured slower (below), not compiled by default
  if (attr_only) {
    hipError_t e = dwt_dk<NT, 2, false>(m, st, true);
    return e == hipSuccess ? dwt_dk<NT, 2, true>(m, st, true) : e;
  }
  if (two) return wl ? dwt_dk<NT, 2, true>(m, st, false) : dwt_dk<NT, 2, false>(m, st, false);
#endif
  (void)two;
  if (attr_only) {
    hipError_t e = dwt_dk<NT, 1, false>(m, st, true);
    return e == hipSuccess ? dwt_dk<NT, 1, true>(m, st, true) : e;
  }
  return wl ? dwt_dk<NT, 1, true>(m, st, false) : dwt_dk<NT, 1, false>(m, st, false);
}

static hipError_t dwt_any(YlConvMulti& m, int NT, hipStream_t st, bool two, bool wl, bool attr_only) {
  hipError_t e = hipSuccess;
  if (attr_only || NT == 1) e = dwt_nt<1>(m, st, two, wl, attr_only);
  if (attr_only ? e == hipSuccess : NT == 2) e = dwt_nt<2>(m, st, two, wl, attr_only);
  if (attr_only ? e == hipSuccess : NT == 3) e = dwt_nt<3>(m, st, two, wl, attr_only);
  if (attr_only ? e == hipSuccess : NT == 4) e = dwt_nt<4>(m, st, two, wl, attr_only);
  if (attr_only ? e == hipSuccess : NT == 6) e = dwt_nt<6>(m, st, two, wl, attr_only);
  return e;
}

// depthwise (3x3 / 5x5, stride 1 / 2) -> 1x1 with N % 4 == 0, N <= 96 (all n-tiles in one wave), OH % 4 == 0,
// OW % 4 == 0.  hipErrorNotSupported otherwise (yl_conv_dwh_kernel then runs the layer).
hipError_t yl_launch_conv_dwt(YlConvMulti& m, hipStream_t st) {
  const YlConvP& p = m.p[0];
  if (p.dw_k == 0 || (p.N & 3) || p.NTtot > 6 || p.dec_boxes || p.C1 > 0) return hipErrorNotSupported;
  // Per-layer measurements (edge_n, B = 64, us; dwh = yl_conv_dwh_kernel, <mt><wl> = this kernel):
  //   K=N=96 dw3 80x80:  dwh 143 | 10: 163 | 11: 137 | 20: 147 | 21: 141        K=96 N=48 dw3: dwh 32.9 | 11: 29.2
  //   K=256 N=64 dw5:    dwh 49.7 | 10: 35.1 | 11: 49.3 | 20: 34.9             K=32 N=96 dw5 s2: dwh 76 | 11: 70
  // -> one m-tile per wave (the 4x8 tile needs ~190 VGPRs: 2 waves per SIMD), LDS weight image while it is <= 36 KB.
  // ("dev_select" bit 4 disables the kernel: developer A/B)
  if (p.dev & YL_DEV_DWT_OFF) return hipErrorNotSupported;
  bool two = false, wl = p.KB * p.NTtot <= 36;
  for (int k = 0; k < m.n; ++k) {
    if ((m.p[k].OH & 3) || (m.p[k].OW & 3)) return hipErrorNotSupported;
    if ((size_t)m.p[k].B * m.p[k].H * m.p[k].W * m.p[k].Cin * sizeof(yl_act_t) >= ((size_t)1 << 31)) return hipErrorNotSupported;   // 32-bit byte offsets
    if (m.p[k].OW & 7) two = false;
  }
  const int nts[5] = {1, 2, 3, 4, 6};
  int NT = 6;
  for (int i = 0; i < 5; ++i) if (nts[i] >= p.NTtot) { NT = nts[i]; break; }
  // split-K: chosen by the layer's SHAPE only (batch-invariant results); "dev_select" bit 10 keeps the one-wave-per-tile form
  if (m.n == 1 && NT == 4 && !wl && p.KB >= 8 && p.OH * p.OW <= 400 && !(p.dev & YL_DEV_DWT_NOSPLIT)) {
    const hipError_t e = dwt_splitk(m, st, false);
    if (e != hipErrorNotSupported) return e;
  }
  return dwt_any(m, NT, st, two, wl, false);
}

// ------------------------------------------------------------------------------------------------
// Dense k x k convolution whose weights do not fit LDS (yololite_m's FPN: 3x3, 328 -> 328 channels = 3.9 MB packed,
// 16.3 of the model's ~24 GMAC per image; model_v2.py:15-22,125-127) as an implicit GEMM with the weight stream
// DOUBLE-BUFFERED through LDS.  yl_conv_mfma_kernel streams the same weights in 48 KiB chunks between two
// workgroup barriers each (fill, then use): while a chunk is fetched the workgroup's four waves issue no MFMA,
// and with 21 n-tiles split 8 + 8 + 5 an eighth of the issued MFMAs multiplies zero padding -- 75 TFLOP/s = 0.48 of
// the fp32 MFMA peak.  Here: NT = 7 n-tiles per workgroup (21 = 3 x 7, no padding), chunks of CH k-steps in two LDS
// buffers, the asynchronous global->LDS copies of chunk c+1 issued before the MFMAs of chunk c (ONE barrier per
// chunk, which is also where the copies are waited for), the chunk pipeline running on across tile boundaries.
// Same transposed GEMM, k order and epilogues as yl_conv_mfma_kernel: bit-identical results.
template <int NT, int MT, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : 4) void yl_conv_kxk_kernel(YlConvP p) {
  extern __shared__ __attribute__((aligned(16))) float yl_clds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  const int TK = p.TK, KB = p.KB, K = p.k, NTtot = p.NTtot;
  const int Cin = p.Cin, H = p.H, W = p.W, stride = p.stride, pad_t = p.pad_t, pad_l = p.pad_l, sh = p.in_shift;
  const int ohw = p.OH * p.OW, OW = p.OW, M = p.M;
  const yl_act_t* const xin = p.x;
  const long zdelta = p.zeros - p.x;
  f32x4* wl = reinterpret_cast<f32x4*>(yl_clds);            // [2][3][NT][64] float4
  const f32x4* wg = reinterpret_cast<const f32x4*>(p.wp);
  const int NC = 3 * KB;                                     // chunks (3 taps each) per (m-tile, n-group) item
  const int G = NTtot / NT;                                  // n-groups of NT n-tiles (the launcher guarantees NTtot % NT == 0)
  // Work order.  A workgroup runs ALL its m-tiles for n-group 0, then for group 1, ...: at any time the workgroups
  // of an XCD stream the same 1/G of the weights (1.3 MB of yololite_m's 3.9 MB), which stays in that XCD's 4 MB L2;
  // with the groups spread over gridDim.y every XCD cycled through the whole image and FETCH_SIZE showed each pass
  // coming from HBM/MALL again (20x the algorithmic bytes).  The m-tiles of an XCD (workgroup b -> XCD b % 8) are
  // one contiguous band of the output, so the 3x3 halo rows of neighbouring tiles hit the same L2 as well.
  const int bx = blockIdx.x, gx = gridDim.x;                 // gx % 8 == 0
  const int per = gx >> 3, slot = bx >> 3;
  const int tpx = (p.ntiles + 7) >> 3;
  const int band0 = (bx & 7) * tpx;
  const int band1 = (band0 + tpx) < p.ntiles ? (band0 + tpx) : p.ntiles;
  const int bt = band1 > band0 ? band1 - band0 : 0;          // (n-group, m-tile) items of the band, group-major;
  const int nitems = bt * G;                                 // workgroup `slot` of the XCD takes items slot, slot+per, ...
  const int nmine = slot < nitems ? (nitems - 1 - slot) / per + 1 : 0;
  const long total_chunks = (long)nmine * NC;

  // asynchronous copy of one chunk = the three taps (ky = cc, kx = 0..2) of channel block kb, n-group g, into buffer
  // `buf`: (tap, n-tile) pieces of 1 KiB dealt round-robin to the waves.  The packed image is tap-major (k-step
  // tap * KB + kb, shared with yl_conv_mfma_kernel); this kernel walks it channel-block-major, see below.
  auto load_chunk = [&](int g, int kb, int cc, int buf) {
    for (int i = wave; i < 3 * NT; i += NW) {
      const int j = i / NT, nt = i - j * NT;
      f32x4* dst = wl + ((size_t)buf * 3 * NT + i) * 64;
      yl_glds16(wg + ((size_t)((cc * 3 + j) * KB + kb) * NTtot + g * NT + nt) * 64 + lane, dst);
    }
  };
  if (total_chunks > 0) load_chunk(slot / bt, 0, 0, 0);
  long gchunk = 0;                                            // chunks consumed so far (buffer = gchunk & 1)
  __syncthreads();
  const bool pre_add = (p.res || p.up) && p.act == YL_ACT_NONE;
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;

  for (int wi = 0; wi < nmine; ++wi) {
    const int item = slot + wi * per;
    const int g = item / bt;
    const int nt0 = g * NT;
    const int tile = band0 + item - g * bt;
    YlPix px[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      size_t lin = ((size_t)tile * NW + wave) * (MT * 16) + mt * 16 + pl;
      px[mt].valid = lin < (size_t)M;
      if (!px[mt].valid) lin = (size_t)M - 1;
      px[mt].lin = lin;
      const int b = (int)(lin / ohw);
      const int rem = (int)(lin - (size_t)b * ohw);
      px[mt].b = b;
      px[mt].oy = rem / OW;
      px[mt].ox = rem - px[mt].oy * OW;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pre_add) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const size_t obase = px[mt].lin * p.N;
        size_t up_off = 0;
        if (p.up) {
          const int uy = (px[mt].oy * p.UH) / p.OH, ux = (px[mt].ox * p.UW) / p.OW;
          up_off = (((size_t)px[mt].b * p.UH + uy) * p.UW + ux) * p.N;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int n = (nt0 + nt) * 16 + 4 * kq;
          if (n < p.N) {
            if (p.res) acc[mt][nt] = yl_ld4(p.res + obase + n);
            if (p.up) acc[mt][nt] += yl_ld4(p.up + up_off + n);
          }
        }
      }
    }
    // K order: channel block outer, the nine taps inner.  The taps of one 16-channel block read a 3 x 18 pixel x 64 B
    // window per wave, which the 9 consecutive k-steps find in L1 / L2; tap-major (the order of yl_conv_mfma_kernel)
    // puts KB = 21 k-steps of other channels between two touches of a line, more than L1 and the XCD's L2 hold, and
    // FETCH_SIZE showed every tap of every pixel coming from HBM / MALL again: 7.5 GB per launch for a 269 MB input.
    // Per lane: nine tap pointers (the zero buffer where a tap falls outside the image, marked in `inb`).
    const yl_act_t* tp[MT][9];
    unsigned inb[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      inb[mt] = 0;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int iy = px[mt].oy * stride - pad_t + tap / 3, ix = px[mt].ox * stride - pad_l + tap % 3;
        const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const long off = in ? ((((long)px[mt].b * (H >> sh) + (iy >> sh)) * (W >> sh) + (ix >> sh)) * Cin + 4 * kq) : zdelta;
        tp[mt][tap] = xin + off;
        inb[mt] |= in ? (1u << tap) : 0u;
      }
    }
    auto fetch = [&](f32x4 (&dst)[MT], int kb, int tap) {
      const bool tail = kb * 16 + 4 * kq >= Cin;                  // channel tail of the last block: zeros
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const yl_act_t* q = tp[mt][tap] + (((inb[mt] >> tap) & 1u) ? kb * 16 : 0);
        dst[mt] = yl_ld4(tail ? xin + zdelta : q);
      }
    };
    f32x4 xq[MT];
    fetch(xq, 0, 0);
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int cc = 0; cc < 3; ++cc, ++gchunk) {
        const int buf = (int)(gchunk & 1);
        // next chunk of the stream (this item's, or the first one of the workgroup's next item) into the other buffer
        if (gchunk + 1 < total_chunks) {
          if (cc < 2) load_chunk(g, kb, cc + 1, buf ^ 1);
          else if (kb + 1 < KB) load_chunk(g, kb + 1, 0, buf ^ 1);
          else load_chunk((item + per) / bt, 0, 0, buf ^ 1);
        }
        const f32x4* wb = wl + (size_t)buf * 3 * NT * 64 + lane;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int tap = cc * 3 + j;
          f32x4 xn[MT];
          if (tap < 8) fetch(xn, kb, tap + 1);
          else if (kb + 1 < KB) fetch(xn, kb + 1, 0);
          else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xn[mt] = xq[mt];
          }
          f32x4 wq[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) wq[nt] = wb[(j * NT + nt) * 64];
          yl_mma_step<NT, MT>(wq, xq, acc);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) xq[mt] = xn[mt];
        }
        __syncthreads();           // every wave is done with `buf`; the copies into the other buffer have landed
      }
    }
    if (!pre_add && (p.res || p.up || YL_SMOOTH(p.act))) yl_epi_generic<NT, MT>(p, acc, px, nt0, kq);
    else yl_epi_fast<NT, MT>(p, acc, px, nt0, kq, lo, hi, true);
  }
}

template <int NT, int MT, int NW>
static hipError_t kxk_go(const YlConvP& p0, int gy, hipStream_t st, bool attr_only) {
  if (attr_only)
    return hipFuncSetAttribute((const void*)yl_conv_kxk_kernel<NT, MT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  YlConvP p = p0;
  p.ntiles = (int)(((long)p.M + 16 * NW * MT - 1) / (16 * NW * MT));
  const size_t lds = (size_t)2 * 3 * NT * 1024;
  static int res = 0;
  if (!res) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)yl_conv_kxk_kernel<NT, MT, NW>, NW * 64, lds) != hipSuccess || nb < 1) nb = 1;
    if (nb > 4) nb = 4;
    res = nb * YL_NUM_CU;
  }
  int gx = res & ~7;
  while (gx > 8 && gx - 8 >= p.ntiles * gy) gx -= 8;       // gy = n-groups: (m-tile, n-group) items
  hipLaunchKernelGGL((yl_conv_kxk_kernel<NT, MT, NW>), dim3(gx), dim3(NW * 64), lds, st, p);
  return hipGetLastError();
}

// dense 3x3 layers with N % 4 == 0 whose weight image exceeds the LDS budget and whose n-tile count is a multiple of 7
// (yololite_m's 328-channel FPN) or exactly 4 (the 64-channel prototype branch of the seg head: two m-tiles per wave
// so that a weight fragment read feeds 8 MFMAs).  hipErrorNotSupported otherwise (yl_conv_mfma_kernel runs the layer).
hipError_t yl_launch_conv_kxk(const YlConvP& p, hipStream_t st) {
  if (p.k != 3 || p.dw_k > 0 || (p.N & 3) || p.dec_boxes || p.C1 > 0 || p.w3p) return hipErrorNotSupported;
  static const int nwtab[4] = {0, 4, 8, -1};                  // developer A/B ("dev_select" bits 7-9)
  const int NWsel = nwtab[YL_DEV_KXK_NW(p.dev)];
  const int MTsel = (p.dev & YL_DEV_KXK_MT2) ? 2 : 1;
  if (p.NTtot == 4) {
    if ((size_t)p.TK * 4 * 1024 <= 96 * 1024 || NWsel < 0) return hipErrorNotSupported;
    return NWsel == 8 ? kxk_go<4, 1, 8>(p, 1, st, false) : kxk_go<4, 2, 4>(p, 1, st, false);
  }
  if (p.NTtot % 7 != 0) return hipErrorNotSupported;
  if ((size_t)p.TK * 7 * 1024 <= 96 * 1024) return hipErrorNotSupported;      // small enough to stay resident: other kernel
  // 8 waves per workgroup share each weight chunk (half the LDS-DMA issue work per MFMA: 107 -> 114 TFLOP/s on
  // yololite_m's 80x80 level) when there are enough 128-pixel items to keep the tail short; 4 otherwise
  const int gy = p.NTtot / 7;
  const bool eight = NWsel ? NWsel == 8 : ((long)p.M / 128) * gy >= 2 * 2 * YL_NUM_CU;
  if (eight) return kxk_go<7, 1, 8>(p, gy, st, false);
  return MTsel == 1 ? kxk_go<7, 1, 4>(p, gy, st, false) : kxk_go<7, 2, 4>(p, gy, st, false);
}

// ------------------------------------------------------------------------------------------------
// Plain 1x1 convolution with MANY channels on both sides (round 3): yololite_m's EfficientNet-Lite blocks at 40x40 /
// 20x20 -- conv_pw 120->720, 208->1248, conv_pwl 720->120, 1248->208, ... (model_v2.py:94-100: timm `ir` blocks) --
// 3.5 of the model's 18.6 ms at 45-63 TFLOP/s through yl_conv_pwt_kernel.  That kernel gives every wave 16 pixels x
// <= 4 n-tiles with BOTH operands straight from L1/L2: five 1-KiB fragment loads per 16 MFMAs, ~40 B/clk per CU at the
// MFMA rate -- it is bound by the vector-memory path, not by the matrix pipe.  Here, as in yl_conv_kxk_kernel: NT
// n-tiles (6..13, an exact divisor of the layer's n-tile count where one exists) per (m-tile, n-group) item, the
// weight stream double-buffered through LDS in chunks of two k-steps shared by the workgroup's NW waves (asynchronous
// copies of chunk c+1 under the MFMAs of chunk c, ONE barrier per chunk, the pipeline running on across items), the
// activation fragment of the next k-step requested before the MFMAs of this one; a wave reads its A fragments from
// LDS in groups of <= 7 (28 VGPRs).  Per 16 pixels and k-step: one L1/L2 load + NT LDS reads for 4 NT MFMAs.  Same k
// order (channel blocks ascending, four sub-steps each) and epilogues as yl_conv_pwt_kernel: bit-identical results.
// DEC (NT == 6, one n-group): the head output under yl_predict -- a wave holds whole rows of <= 96 logits and runs the decode
// epilogue of yl_conv_pwt_kernel on them (yololite_m's 328 -> 85 outputs: 21 k-blocks of weights through LDS once per 64 / 128 pixels
// instead of once per 32 through the vector-memory path).
// NT2 > 0 (with DEC): NT2 more n-tiles from a SECOND weight image (YlConvP::w3p / b3 / C3: the mask coefficients of a segmentation
// head, model_v2.py head output split into 5 + C detection columns and NM coefficient columns) ride in the same launch -- the wave's
// rows are read once instead of once per part; the extra columns are stored plain into the level rows (p.out, p.ldo).
template <int NT, int NW, bool SC = false, bool DEC = false, int NT2 = 0>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : 4) void yl_conv_pws_kernel(YlConvP p) {
  constexpr int CH = 2;                                      // k-steps per weight chunk
  constexpr int NTA = NT + NT2;                              // n-tiles a wave accumulates
  extern __shared__ __attribute__((aligned(16))) float yl_clds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  const int KB = p.KB, NTtot = p.NTtot, Cin = p.Cin, M = p.M;
  const yl_act_t* const xin = p.x;
  f32x4* wl = reinterpret_cast<f32x4*>(yl_clds);            // [2][CH][NTA][64] float4
  const f32x4* wg = reinterpret_cast<const f32x4*>(p.wp);   // [KB][NTtot][64] float4
  const int NC = (KB + CH - 1) / CH;                         // chunks per item
  const int G = (NTtot + NT - 1) / NT;                       // n-groups (the last one may be partial: clamped reads, no stores)
  // work order as in yl_conv_kxk_kernel: XCD x (workgroup b -> XCD b % 8) owns a contiguous band of m-tiles and runs
  // its (n-group, m-tile) items group-major, so that at any time an XCD streams 1/G of the weights out of its own L2
  const int bx = blockIdx.x, gx = gridDim.x;                 // gx % 8 == 0
  const int per = gx >> 3, slot = bx >> 3;
  const int tpx = (p.ntiles + 7) >> 3;
  const int band0 = (bx & 7) * tpx;
  const int band1 = (band0 + tpx) < p.ntiles ? (band0 + tpx) : p.ntiles;
  const int bt = band1 > band0 ? band1 - band0 : 0;
  const int nitems = bt * G;
  const int nmine = slot < nitems ? (nitems - 1 - slot) / per + 1 : 0;
  const long total_chunks = (long)nmine * NC;
  // asynchronous copy of chunk `c` (k-steps c*CH .. c*CH+CH-1, clamped) of n-group g into buffer `buf`
  auto load_chunk = [&](int g, int c, int buf) {
    for (int i = wave; i < CH * NTA; i += NW) {
      const int j = i / NTA, nt = i - j * NTA;
      int kb = c * CH + j;
      kb = kb < KB ? kb : KB - 1;
      if (NT2 > 0 && nt >= NT) {                              // (wave-uniform) second image: [KB][NT2][64]
        yl_glds16(reinterpret_cast<const f32x4*>(p.w3p) + ((size_t)kb * NT2 + (nt - NT)) * 64 + lane, wl + ((size_t)buf * CH * NTA + i) * 64);
        continue;
      }
      int ntg = g * NT + nt;
      ntg = ntg < NTtot ? ntg : NTtot - 1;
      yl_glds16(wg + ((size_t)kb * NTtot + ntg) * 64 + lane, wl + ((size_t)buf * CH * NTA + i) * 64);
    }
  };
  if (total_chunks > 0) load_chunk(slot / bt, 0, 0);
  long gchunk = 0;                                            // chunks consumed so far (buffer = gchunk & 1)
  __syncthreads();
  const bool pre_add = (p.res || p.up) && p.act == YL_ACT_NONE;
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const int ohw = p.OH * p.OW;

  for (int wi = 0; wi < nmine; ++wi) {
    const int item = slot + wi * per;
    const int g = item / bt;
    const int nt0 = g * NT;
    const int tile = band0 + item - g * bt;
    YlPix px[1];
    {
      size_t lin = ((size_t)tile * NW + wave) * 16 + pl;
      px[0].valid = lin < (size_t)M;
      if (!px[0].valid) lin = (size_t)M - 1;
      px[0].lin = lin;
      px[0].b = 0; px[0].oy = 0; px[0].ox = 0;
      if (SC || DEC || p.up) {                               // only the upsample-add / decode epilogues / the gate need coordinates
        const int b = (int)(lin / ohw);
        const int rem = (int)(lin - (size_t)b * ohw);
        px[0].b = b;
        px[0].oy = rem / p.OW;
        px[0].ox = rem - px[0].oy * p.OW;
      }
    }
    const yl_act_t* xrow = xin + px[0].lin * Cin + 4 * kq;
    const float* srow = SC ? p.scale + (size_t)px[0].b * Cin + 4 * kq : nullptr;    // squeeze-excite gate of the pixel's image
    f32x4 acc[1][NTA];
#pragma unroll
    for (int nt = 0; nt < NTA; ++nt) acc[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pre_add) {
      const size_t obase = px[0].lin * p.N;
      size_t up_off = 0;
      if (p.up) {
        const int uy = (px[0].oy * p.UH) / p.OH, ux = (px[0].ox * p.UW) / p.OW;
        up_off = (((size_t)px[0].b * p.UH + uy) * p.UW + ux) * p.N;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = (nt0 + nt) * 16 + 4 * kq;
        if (n < p.N) {
          if (p.res) acc[0][nt] = yl_ld4(p.res + obase + n);
          if (p.up) acc[0][nt] += yl_ld4(p.up + up_off + n);
        }
      }
    }
    auto fetch = [&](int kb) {
      const bool ok = kb * 16 + 4 * kq < Cin;                 // channel tail of the last block: zeros
      f32x4 v = yl_ld4(ok ? xrow + kb * 16 : p.zeros);
      if (SC) v *= yl_ld4(ok ? srow + kb * 16 : reinterpret_cast<const float*>(p.zeros));     // x * gate (one rounding), then the GEMM
      return v;
    };
    // activation fragments PF k-steps ahead (a ring with compile-time slots: the chunk loop is unrolled by two).  One step ahead
    // was 1 KB in flight per wave -- 16-32 KB per CU, which at the ~2 us of a loaded HBM / MALL round trip is ~2 TB/s over the
    // chip: the 328 -> 85 head outputs and the 528 -> 88 projections sat on that, not on the matrix pipe.
    constexpr int PF = 4;
    f32x4 xr[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) xr[i] = fetch(i < KB ? i : KB - 1);
    auto chunk = [&](auto par, int c) {
      constexpr int P = decltype(par)::value;
      const int buf = (int)(gchunk & 1);
      // next chunk of the stream (this item's, or the first one of the workgroup's next item) into the other buffer
      if (gchunk + 1 < total_chunks) {
        if (c + 1 < NC) load_chunk(g, c + 1, buf ^ 1);
        else load_chunk((item + per) / bt, 0, buf ^ 1);
      }
      const f32x4* wb = wl + (size_t)buf * CH * NTA * 64 + lane;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int kb = c * CH + j;
        if (kb < KB) {                                         // (workgroup-uniform) odd KB: the last chunk is half empty
          const int slot = (P * CH + j) % PF;
          constexpr int H0 = NTA > 7 ? (NTA + 1) / 2 : NTA;    // A fragments in two groups: <= 28 VGPRs of them live
          constexpr int H1 = NTA - H0;
          f32x4 xs[1] = {xr[slot]};
          xr[slot] = fetch(kb + PF < KB ? kb + PF : KB - 1);
          {
            f32x4 wq[H0], a0[1][H0];
#pragma unroll
            for (int nt = 0; nt < H0; ++nt) { wq[nt] = wb[(j * NTA + nt) * 64]; a0[0][nt] = acc[0][nt]; }
            yl_mma_step<H0, 1>(wq, xs, a0);
#pragma unroll
            for (int nt = 0; nt < H0; ++nt) acc[0][nt] = a0[0][nt];
          }
          if (H1 > 0) {
            constexpr int H1s = H1 > 0 ? H1 : 1;
            f32x4 wq[H1s], a1[1][H1s];
#pragma unroll
            for (int nt = 0; nt < H1; ++nt) { wq[nt] = wb[(j * NTA + H0 + nt) * 64]; a1[0][nt] = acc[0][H0 + nt]; }
            yl_mma_step<H1s, 1>(wq, xs, a1);
#pragma unroll
            for (int nt = 0; nt < H1; ++nt) acc[0][H0 + nt] = a1[0][nt];
          }
        }
      }
      __syncthreads();             // every wave is done with `buf`; the copies into the other buffer have landed
      ++gchunk;
    };
    for (int c = 0; c < NC; c += 2) {
      chunk(std::integral_constant<int, 0>{}, c);
      if (c + 1 < NC) chunk(std::integral_constant<int, 1>{}, c + 1);
    }
    if (DEC) {                                                // (one n-group: nt0 == 0)
      if (NT2 > 0) {
        constexpr int N2 = NT2 > 0 ? NT2 : 1;
        f32x4 ad[1][NT], am[1][N2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ad[0][nt] = acc[0][nt];
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) am[0][nt] = acc[0][NT + nt];
        YlConvP pm = p;                                       // the second part: plain columns into the level rows
        pm.bias = p.b3; pm.N = p.C3;
        yl_epi_fast<N2, 1>(pm, am, px, 0, kq, lo, hi, true);
        yl_epi_decode<NT, 1, false, true>(p, ad, px, 0, kq, lane);
      } else {
        yl_epi_decode<NT, 1, false, true>(p, *reinterpret_cast<f32x4 (*)[1][NT]>(&acc), px, 0, kq, lane);
      }
      continue;
    }
    if (!pre_add && (p.res || p.up || YL_SMOOTH(p.act))) yl_epi_generic<NTA, 1>(p, acc, px, nt0, kq);
    else yl_epi_fast<NTA, 1>(p, acc, px, nt0, kq, lo, hi, true);
  }
}

template <int NT, int NW>
static hipError_t pws_go(const YlConvP& p0, hipStream_t st, bool attr_only) {
  if (attr_only) {
    hipError_t e = hipFuncSetAttribute((const void*)yl_conv_pws_kernel<NT, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
    if (NT == 6) {
      e = hipFuncSetAttribute((const void*)yl_conv_pws_kernel<NT == 6 ? 6 : 6, NW, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      if (e != hipSuccess) return e;
      e = hipFuncSetAttribute((const void*)yl_conv_pws_kernel<NT == 6 ? 6 : 6, NW, false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute((const void*)yl_conv_pws_kernel<NT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  }
  YlConvP p = p0;
  p.ntiles = (int)(((long)p.M + 16 * NW - 1) / (16 * NW));
  const size_t lds = (size_t)2 * 2 * NT * 1024;
  int gx = yl_resident_blocks_n(yl_conv_pws_kernel<NT, NW>, NW * 64, lds) & ~7;
  const int gy = (p.NTtot + NT - 1) / NT;
  while (gx > 8 && gx - 8 >= p.ntiles * gy) gx -= 8;
  if (p.dec_boxes) {
    if (NT != 6 || p.scale) return hipErrorNotSupported;
    if (p.w3p) {                                              // + the mask coefficients (32 columns) from the second weight image
      const size_t lds2 = (size_t)2 * 2 * 8 * 1024;
      int gx2 = yl_resident_blocks_n(yl_conv_pws_kernel<6, NW, false, true, 2>, NW * 64, lds2) & ~7;
      while (gx2 > 8 && gx2 - 8 >= p.ntiles * gy) gx2 -= 8;
      hipLaunchKernelGGL((yl_conv_pws_kernel<6, NW, false, true, 2>), dim3(gx2), dim3(NW * 64), lds2, st, p);
    } else
      hipLaunchKernelGGL((yl_conv_pws_kernel<6, NW, false, true>), dim3(gx), dim3(NW * 64), lds, st, p);
  } else if (p.scale) hipLaunchKernelGGL((yl_conv_pws_kernel<NT, NW, true>), dim3(gx), dim3(NW * 64), lds, st, p);
  else hipLaunchKernelGGL((yl_conv_pws_kernel<NT, NW>), dim3(gx), dim3(NW * 64), lds, st, p);
  return hipGetLastError();
}

template <int NT>
static hipError_t pws_nw(const YlConvP& p, hipStream_t st, bool eight, bool attr_only) {
  if (attr_only) {
    const hipError_t e = pws_go<NT, 4>(p, st, true);
    return e != hipSuccess ? e : pws_go<NT, 8>(p, st, true);
  }
  return eight ? pws_go<NT, 8>(p, st, false) : pws_go<NT, 4>(p, st, false);
}

// plain 1x1 stride-1 layers (N % 4 == 0, no decode epilogue) with enough channels on both sides that the weight
// stream pays: K >= 80 and >= 6 n-tiles.  hipErrorNotSupported otherwise (yl_conv_pwt_kernel runs the layer).
hipError_t yl_launch_conv_pws(const YlConvP& p, hipStream_t st) {
  const bool dec = p.dec_boxes && !p.dec_raw;                  // head output under yl_predict: rows of <= 96 logits, one n-group
  if (p.k != 1 || p.stride != 1 || p.dw_k > 0 || p.C1 > 0 || p.in_shift || (p.dec_boxes && !dec)) return hipErrorNotSupported;
  if (dec ? (p.NTtot != 6 || p.scale || p.res || p.up || (p.w3p && (p.C3 != 32 || YL_SMOOTH(p.act)))) : ((p.N & 3) != 0 || p.w3p)) return hipErrorNotSupported;
  if ((p.dev & YL_DEV_PWS_OFF) || p.KB < 5 || p.NTtot < 6) return hipErrorNotSupported;     // (dev: A/B runs)
  // n-tiles per item, from {6..13}: the makespan of the launch in MFMA units -- (16-pixel x n-group) wave items dealt
  // to 1024 SIMDs, each NT x KB x 4 MFMAs long -- with a penalty when fewer than 1.5 waves per SIMD exist (one wave
  // alone cannot keep a matrix pipe busy); ties go to the larger NT (fewer passes over the activations).  E.g. 1248 ->
  // 208 at 20x20 x 32 images (13 n-tiles, 800 pixel tiles): NT = 7 in two groups (116 -> 90 us), not 13 in one.
  static const int cand[6] = {13, 11, 9, 8, 7, 6};
  int NT = 0;
  double best = 1e30;
  long best_wi = 0;
  const long mt16 = ((long)p.M + 15) / 16;
  for (int i = dec ? 5 : 0; i < 6; ++i) {
    const long wi = mt16 * ((p.NTtot + cand[i] - 1) / cand[i]);
    const double sc = (double)((wi + 1023) / 1024) * cand[i] * (wi < 1536 ? 1.25 : 1.0);
    if (sc < best) { best = sc; NT = cand[i]; best_wi = wi; }
  }
  if (best_wi < 1536) return hipErrorNotSupported;      // too few items for workgroup-shared weights: wave-autonomous kernel
  const int gy = (p.NTtot + NT - 1) / NT;
  const bool eight = ((long)p.M / 128) * gy >= 2 * 2 * YL_NUM_CU;          // enough 128-pixel items for 8-wave workgroups
  switch (NT) {
    case 6: return pws_nw<6>(p, st, eight, false);
    case 7: return pws_nw<7>(p, st, eight, false);
    case 8: return pws_nw<8>(p, st, eight, false);
    case 9: return pws_nw<9>(p, st, eight, false);
    case 11: return pws_nw<11>(p, st, eight, false);
    default: return pws_nw<13>(p, st, eight, false);
  }
}

// ------------------------------------------------------------------------------------------------
// Whole EfficientNet-style inverted-residual block in ONE launch (round 3):  1x1 expand (+BN+act) -> depthwise DK x DK,
// stride 1 or 2, TF-SAME or symmetric padding (+BN+act) -> 1x1 project (+BN) (+residual) -- timm's `ir` blocks behind
// model_v2.py:94-100.  Built for the EARLY blocks of yololite_m (tf_efficientnet_lite2: 16->96->24 at 320x320, 24->144->24 at
// 160x160, 48->288->48 at 80x80, ...), where the 6x expanded tensor is the largest tensor of the network (1.26 GB at B = 32)
// and the two-launch form (conv_pw, then depthwise + conv_pwl) spends its time writing and re-reading it at 3-4 TB/s:
// 2.7 of the model's 17.6 ms.  Here the expanded tensor exists only as one 16-channel SLAB of the workgroup's halo region
// in LDS.
//   workgroup tile  4*RBN x 4*MT*CBN output pixels, RBN x CBN waves (2 x 2, or 1 x 5 for 20 x 20 grids), wave w owns the
//                   4 x 4*MT block (w / CBN, w % CBN)
//   per slab kb     E: the halo region ((TH-1)*DS+DK rows x (TW-1)*DS+DK columns of expanded pixels, 16-pixel m-tiles
//                      dealt to the waves) = act(bias + Wexp[kb] . x) by MFMA from the block input held in registers
//                      for the whole tile, zero outside the image (the depthwise conv pads the EXPANDED tensor) -> LDS;
//                   barrier (ONE per slab: slabs and projection weights are double-buffered, the counters run on
//                      across tiles);
//                   D: B fragment of the lane's pixel = act(bias + sum of taps), taps and weights from LDS, in the tap
//                      order of yl_conv_dwt_kernel;  P: NT x 4 MFMAs per m-tile against the projection weights of
//                      the slab (asynchronous global -> LDS copy issued one slab ahead).
// The old yl_uib_kernel (yl_conv.hip, MobileNetV4 blocks) recomputes the expansion on a (3+DK)^2 halo per 4x4 tile and
// WAVE: 2.25-4x the expansion MFMAs; the workgroup-level halo here costs 1.2x (stride 2) to 1.9x (5x5 stride 1).  Same
// k orders (input channel blocks ascending; slabs = the projection's k blocks ascending; taps (dy,dx)) and epilogues as
// the two-launch form: bit-identical results.
// (Round 3, second half: a wave-autonomous form of the 48 -> 96 -> 48 blocks on an LDS-DMA-staged 6x6 input patch -- the
// recipe of yl_conv_s2c_kernel, no workgroup barrier, 288 MFMAs per 16 pixels instead of 198 -- was built and was
// bit-identical: 49.1 us per B = 64 launch against 46.2 us here, and 38.4k against 40.0k images/s on the two-stream
// headline (118 KB of LDS per CU also keeps the other chunk's kernels off the CU).  Not kept.  Likewise a START-depthwise
// form of this kernel for edge_n blocks.2.0 (dw5 -> 32->96 -> dw5 s2 -> 96->48 as one launch: the expansion's B fragments
// computed per halo pixel from 25 x 2 float4 taps read straight from L1/L2, bit-identical, the 157 MB expanded tensor
// never written): 35.4k against 40.0k images/s -- 300 dependent tap loads per lane and tile at two workgroups per CU;
// it would need the block input staged in LDS (68 KB per 8x8 tile).  Not kept.)
template <int KBI /*ceil(C1/16)*/, int NT, int DK, int DS, int MT, int RBN /*wave rows*/, int CBN /*wave columns*/>
__global__ __launch_bounds__(RBN * CBN * 64, (DK == 3 && DS * MT <= 2 && KBI <= 3 && NT <= 3) ? 3 : 2) void yl_ir_kernel(YlConvP p) {
  constexpr int NWV = RBN * CBN, NTH = NWV * 64;                   // waves / threads per workgroup
  constexpr int TH = 4 * RBN, TW = 4 * MT * CBN;                   // workgroup tile: every wave a 4 x 4 MT block
  constexpr int HH = (TH - 1) * DS + DK, HW = (TW - 1) * DS + DK, HN = HH * HW;
  constexpr int HMT = (HN + 15) / 16, HMW = (HMT + NWV - 1) / NWV; // halo m-tiles: all, per wave
  constexpr int PITCHF = ((HW * 16 + 7) / 64) * 64 + 56;           // slab row pitch in floats (see yl_conv_dwh_kernel)
  constexpr int SLAB = HH * PITCHF;
  extern __shared__ __attribute__((aligned(16))) float yl_clds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  const int Cmid = p.Cin, KB = p.KB, NTtot = p.NTtot, C1 = p.C1, H = p.H, W = p.W, OH = p.OH, OW = p.OW, N = p.N;
  float* slab = yl_clds;                                           // [2][SLAB]
  f32x4* wpl = reinterpret_cast<f32x4*>(yl_clds + 2 * SLAB);       // [2][NT][64] float4: projection weights of a slab
  float* dwl = yl_clds + 2 * SLAB + 2 * NT * 256;                  // [KB][DK*DK + 1][16]: taps + dw bias of a slab's channels (zeros beyond Cmid):
                                                                   // k-block-major, so that a lane's tap reads of a slab are ONE address + immediates
  float* b2l = dwl + (size_t)KB * (DK * DK + 1) * 16;              // [KB*16] expansion bias
  const f32x4* wg = reinterpret_cast<const f32x4*>(p.wp);          // projection [KB][NTtot][64]
  const f32x4* w2g = reinterpret_cast<const f32x4*>(p.w2p);        // expansion [KBI][KB][64]
  const yl_act_t* const xin = p.x;
  const yl_act_t* const up = p.up;                                     // FPN lateral + smooth pair: addend of the expansion
  {
    constexpr int T1 = DK * DK + 1;
    for (int i = tid; i < KB * T1 * 16; i += NTH) {
      const int kb = i / (T1 * 16), r = i - kb * (T1 * 16), t = r >> 4, ch = kb * 16 + (r & 15);
      dwl[i] = ch < Cmid ? (t < DK * DK ? p.dw_w[(size_t)t * Cmid + ch] : (p.dw_b ? p.dw_b[ch] : 0.0f)) : 0.0f;
    }
    for (int i = tid; i < KB * 16; i += NTH) b2l[i] = p.b2[i];
  }
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float dlo = (p.dw_act == YL_ACT_RELU || p.dw_act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float dhi = (p.dw_act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float elo = (p.act2 == YL_ACT_RELU || p.act2 == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float ehi = (p.act2 == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const int dw_act = p.dw_act, act2 = p.act2;
  // lane constants: the lane's halo pixel in each of the wave's halo m-tiles (m = wave + 4 j)
  int h_r[HMW], h_c[HMW], h_lo[HMW];
  bool h_ok[HMW];
#pragma unroll
  for (int j = 0; j < HMW; ++j) {
    const int q = (wave + NWV * j) * 16 + pl;
    h_ok[j] = q < HN;
    const int qq = h_ok[j] ? q : 0;
    h_r[j] = qq / HW;
    h_c[j] = qq - h_r[j] * HW;
    h_lo[j] = h_r[j] * PITCHF + h_c[j] * 16 + 4 * kq;
  }
  const int rb = wave / CBN, cb = wave - rb * CBN;                  // the wave's 4 x 4 MT block of the tile
  const int rbase = ((4 * rb + (pl >> 2)) * DS) * PITCHF + ((cb * 4 * MT + (pl & 3)) * DS) * 16 + 4 * kq;
  const bool pre_add = p.res != nullptr && p.act == YL_ACT_NONE;
  const int twn = OW / TW, thn = OH / TH;
  const int tiles_img = twn * thn;
  const int ntiles = p.B * tiles_img;
  int tile, tend, tstride;
  if ((gridDim.x & 7) == 0) {                                      // XCD bands, see yl_conv_dwt_kernel
    const int tpx = (ntiles + 7) >> 3;
    const int band0 = (blockIdx.x & 7) * tpx;
    tend = (band0 + tpx) < ntiles ? (band0 + tpx) : ntiles;
    tile = band0 + (blockIdx.x >> 3);
    tstride = gridDim.x >> 3;
  } else {
    tile = blockIdx.x; tend = ntiles; tstride = gridDim.x;
  }
  auto load_proj = [&](int kb, int buf) {                           // projection weights of slab kb -> LDS (asynchronous)
    for (int nt = wave; nt < NT; nt += NWV)
      yl_glds16(wg + ((size_t)kb * NTtot + (nt < NTtot ? nt : NTtot - 1)) * 64 + lane, wpl + ((size_t)buf * NT + nt) * 64);
  };
  unsigned gs = 0;                                                  // slabs started by this workgroup (buffer = gs & 1)
  if (tile < tend) load_proj(0, 0);
  __syncthreads();

  for (int wi = 0; tile < tend; tile += tstride, ++wi) {             // (wi: the stamp builds' tile counter)
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int tyi = trem / twn, txi = trem - tyi * twn;
    const int iy0 = tyi * TH * DS - p.dw_pad_t, ix0 = txi * TW * DS - p.dw_pad_l;
    YlPix px[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      px[mt].b = b;
      px[mt].oy = tyi * TH + 4 * rb + (pl >> 2);
      px[mt].ox = txi * TW + cb * 4 * MT + 4 * mt + (pl & 3);
      px[mt].valid = true;
      px[mt].lin = ((size_t)b * OH + px[mt].oy) * OW + px[mt].ox;
    }
    // block input at the lane's halo pixels: B fragments of the expansion GEMM, resident for the tile
    f32x4 xh[HMW][KBI];
    bool h_in[HMW];
    long uoff[HMW];
#pragma unroll
    for (int j = 0; j < HMW; ++j) {
      const int iy = iy0 + h_r[j], ix = ix0 + h_c[j];
      h_in[j] = h_ok[j] && iy >= 0 && iy < H && ix >= 0 && ix < W;
      uoff[j] = 0;
      if (up && h_in[j]) uoff[j] = ((((long)b * p.UH + (iy * p.UH) / H) * p.UW + (ix * p.UW) / W) * Cmid) + 4 * kq;
      const yl_act_t* src = xin + (((size_t)b * H + iy) * W + ix) * C1 + 4 * kq;
#pragma unroll
      for (int kbi = 0; kbi < KBI; ++kbi) {
        const bool ok = h_in[j] && (kbi * 16 + 4 * kq) < C1;
        xh[j][kbi] = yl_ld4(ok ? src + kbi * 16 : p.zeros);
      }
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int n = nt * 16 + 4 * kq;
        if (pre_add && n < N) acc[mt][nt] = yl_ld4(p.res + px[mt].lin * N + n);
      }
    f32x4 we[KBI], wn[KBI];
#pragma unroll
    for (int kbi = 0; kbi < KBI; ++kbi) wn[kbi] = w2g[((size_t)kbi * KB + 0) * 64 + lane];
    // FPN lateral: the addend of slab kb + 1 (the nearest-upsampled coarser level at the lane's halo pixels) is requested
    // one slab ahead like the expansion weights -- requested where it is used, every slab began with a wait for a
    // fragment-shaped load from L2 in front of its first MFMA (it initialises the accumulator)
    // (one register set: m-tile j's addend of slab kb + 1 is requested right behind the MFMAs that consumed slab kb's)
    f32x4 un[HMW];
    auto load_up = [&](int kb, int j) {
      return yl_ld4((h_in[j] && kb * 16 + 4 * kq < Cmid) ? up + uoff[j] + kb * 16 : p.zeros);
    };
    if (up) {
#pragma unroll
      for (int j = 0; j < HMW; ++j) un[j] = load_up(0, j);
    }
    for (int kb = 0; kb < KB; ++kb, ++gs) {
      const int buf = (int)(gs & 1u);
#pragma unroll
      for (int kbi = 0; kbi < KBI; ++kbi) we[kbi] = wn[kbi];
      if (kb + 1 < KB) {
#pragma unroll
        for (int kbi = 0; kbi < KBI; ++kbi) wn[kbi] = w2g[((size_t)kbi * KB + kb + 1) * 64 + lane];
      }
      WINO_STAMP(kb * 5 + 0);
      // ---- E: expansion slab on the wave's halo m-tiles -> LDS
      const f32x4 eb = yl_ld4(b2l + kb * 16 + 4 * kq);
      float* sb = slab + buf * SLAB;
#pragma unroll
      for (int j = 0; j < HMW; ++j) {
        f32x4 e[1][1] = {{{0.f, 0.f, 0.f, 0.f}}};
        if (up) e[0][0] = un[j];   // FPN lateral: the nearest-upsampled coarser level initialises the accumulator (the order
                                   // of the stand-alone lateral conv: (addend + products) + bias)
#pragma unroll
        for (int kbi = 0; kbi < KBI; ++kbi) {
          const f32x4 wq1[1] = {we[kbi]};
          const f32x4 xq1[1] = {xh[j][kbi]};
          yl_mma_step<1, 1>(wq1, xq1, e);
        }
        if (up && kb + 1 < KB) un[j] = load_up(kb + 1, j);
        const f32x4 v = yl_sel4(h_in[j], yl_actc(e[0][0] + eb, act2, elo, ehi));   // zero padding of the EXPANDED tensor
        if (h_ok[j]) *reinterpret_cast<f32x4*>(sb + h_lo[j]) = v;
      }
      WINO_STAMP(kb * 5 + 1);
      __syncthreads();                 // slab `buf` complete; the projection weights of this slab have landed
      WINO_STAMP(kb * 5 + 2);
      if (kb + 1 < KB) load_proj(kb + 1, buf ^ 1);
      else if (tile + tstride < tend) load_proj(0, buf ^ 1);        // first slab of the workgroup's next tile
      // ---- D: depthwise on the slab -> B fragments
      const float* tapw = dwl + kb * ((DK * DK + 1) * 16) + 4 * kq;
      f32x4 xq[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xq[mt] = yl_ld4(tapw + DK * DK * 16);
      auto tap_row = [&](int dy) {
#pragma unroll
        for (int dx = 0; dx < DK; ++dx) {
          const f32x4 w = yl_ld4(tapw + (dy * DK + dx) * 16);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sb + rbase + dy * PITCHF + (dx + 4 * mt * DS) * 16);
            xq[mt].x = fmaf(v.x, w.x, xq[mt].x); xq[mt].y = fmaf(v.y, w.y, xq[mt].y);
            xq[mt].z = fmaf(v.z, w.z, xq[mt].z); xq[mt].w = fmaf(v.w, w.w, xq[mt].w);
          }
        }
      };
#pragma unroll 1
      for (int dy = 0; dy < DK; ++dy) tap_row(dy);                   // one tap row at a time bounds the register footprint
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xq[mt] = yl_actc(xq[mt], dw_act, dlo, dhi);
      WINO_STAMP(kb * 5 + 3);
      // channel tail (c >= Cmid): the packed projection weights of those k slots are zero, no select needed
      // ---- P: projection
      f32x4 wq[NT];
      const f32x4* wrow = wpl + (size_t)buf * NT * 64 + lane;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wq[nt] = wrow[nt * 64];
      yl_mma_step<NT, MT>(wq, xq, acc);
#ifdef YL_WINO_STAMP
      __builtin_amdgcn_sched_barrier(0);
#endif
      WINO_STAMP(kb * 5 + 4);
    }
    WINO_STAMP(KB * 5);
    if (!pre_add && (p.res || YL_SMOOTH(p.act))) yl_epi_generic<NT, MT>(p, acc, px, 0, kq);
    else yl_epi_fast<NT, MT>(p, acc, px, 0, kq, lo, hi, true);
  }
}

static size_t yl_ir_lds(int dk, int ds, int mt, int rbn, int cbn, int nt, int cmid) {
  const int hh = (4 * rbn - 1) * ds + dk, hw = (4 * mt * cbn - 1) * ds + dk;
  const int pitch = ((hw * 16 + 7) / 64) * 64 + 56;
  return ((size_t)2 * hh * pitch + (size_t)2 * nt * 256 + (size_t)((cmid + 15) / 16) * 16 * (dk * dk + 1) + (size_t)((cmid + 15) / 16) * 16) * 4;
}

template <int KBI, int NT, int DK, int DS, int MT, int RBN, int CBN>
static hipError_t ir_go(const YlConvP& p, hipStream_t st, bool attr_only) {
  if (attr_only)
    return hipFuncSetAttribute((const void*)yl_ir_kernel<KBI, NT, DK, DS, MT, RBN, CBN>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  const size_t lds = yl_ir_lds(DK, DS, MT, RBN, CBN, NT, p.Cin);
  const long ntiles = (long)p.B * (p.OH / (4 * RBN)) * (p.OW / (4 * MT * CBN));
  int gx = yl_resident_blocks_n(yl_ir_kernel<KBI, NT, DK, DS, MT, RBN, CBN>, RBN * CBN * 64, lds);
  if (gx > ntiles) gx = (int)ntiles;
  if (gx >= 8) gx &= ~7;
  hipLaunchKernelGGL((yl_ir_kernel<KBI, NT, DK, DS, MT, RBN, CBN>), dim3(gx), dim3(RBN * CBN * 64), lds, st, p);
  return hipGetLastError();
}

// instantiated shapes: (input k-blocks, projection n-tile bucket, dw k, dw stride, m-tiles per wave, wave rows, wave columns).
// Buckets: 2 = 1-2 n-tiles, 3 = 3, 4 = 4, 6 = 5-6.  Workgroup tile = 4*rows x 4*MT*columns output pixels: 2 x 2 waves for
// the grids that are multiples of 8 / 16.  (1 x 5 waves = 4 x 20 pixel tiles for the 20 x 20 stage of edge_n were built
// and measured: 320 workgroup tiles per B = 64 launch, 0.083-0.114 ms per block against 0.055-0.067 ms for the
// two-launch form -- not instantiated.)
#define YL_IR_SHAPES(X)                                                                                              \
  X(1, 2, 3, 2, 1, 2, 2) X(2, 2, 3, 1, 2, 2, 2) X(2, 3, 5, 2, 1, 2, 2) X(3, 3, 5, 1, 2, 2, 2) X(3, 6, 3, 2, 1, 2, 2)    \
  X(3, 3, 3, 1, 1, 2, 2) X(2, 6, 3, 1, 2, 2, 2) X(3, 6, 3, 1, 1, 2, 2)
#define YL_IR_BUCKET_LO(B) ((B) == 2 ? 0 : (B) == 3 ? 2 : (B) == 4 ? 3 : 4)

// fused inverted-residual block (p.C1 > 0).  hipErrorNotSupported: shape not instantiated (yl_uib_kernel or the
// two-launch form handles it -- yl_ir_supported tells the host compiler beforehand)
bool yl_ir_supported(int c1, int cmid, int n, int dk, int ds, int oh, int ow) {
  const int kbi = (c1 + 15) / 16, nt = (n + 15) / 16;
#define YL_IR_CHECK(A, B, C, D, E, R, S)                                                                             \
  if (kbi == A && nt <= B && nt > YL_IR_BUCKET_LO(B) && dk == C && ds == D && (oh % (4 * R)) == 0 &&                    \
      (ow % (4 * E * S)) == 0 && yl_ir_lds(C, D, E, R, S, B, cmid) <= 150 * 1024) return true;
  YL_IR_SHAPES(YL_IR_CHECK)
#undef YL_IR_CHECK
  return false;
}

hipError_t yl_launch_conv_ir(const YlConvP& p, hipStream_t st) {
  if (p.C1 <= 0 || p.k != 1 || p.dw_k == 0 || (p.N & 3)) return hipErrorNotSupported;
  const int kbi = (p.C1 + 15) / 16, nt = p.NTtot;
#define YL_IR_RUN(A, B, C, D, E, R, S)                                                                               \
  if (kbi == A && nt <= B && nt > YL_IR_BUCKET_LO(B) && p.dw_k == C && p.dw_stride == D && (p.OH % (4 * R)) == 0 &&     \
      (p.OW % (4 * E * S)) == 0 && yl_ir_lds(C, D, E, R, S, B, p.Cin) <= 150 * 1024)                                    \
    return ir_go<A, B, C, D, E, R, S>(p, st, false);
  YL_IR_SHAPES(YL_IR_RUN)
#undef YL_IR_RUN
  return hipErrorNotSupported;
}

static hipError_t yl_ir_init() {
  YlConvP q = {};
  hipError_t e = hipSuccess;
#define YL_IR_ATTR(A, B, C, D, E, R, S) if (e == hipSuccess) e = ir_go<A, B, C, D, E, R, S>(q, nullptr, true);
  YL_IR_SHAPES(YL_IR_ATTR)
#undef YL_IR_ATTR
  return e;
}

// ------------------------------------------------------------------------------------------------
// Depthwise 3x3 -> 1x1 convolution whose 1x1 weights do not fit LDS (edge_m's 244-channel and yololite_m's
// 328-channel neck / head blocks, model_v2.py:24-41: 240-430 KB packed).  yl_conv_dwh_kernel cannot hold the image and
// the layer fell to yl_conv_mfma_kernel's streamed mode (48 TFLOP/s: fill / use barriers, 21 n-tiles as 8 + 8 + 5).
// Same machinery as yl_conv_kxk_kernel: NT = 7 or 8 n-tiles per workgroup item, the weight stream double-buffered
// through LDS in chunks of three k-steps with one barrier per chunk, (n-group, m-tile) items dealt group-major in XCD
// bands.  The B operand of k-step kb is the depthwise result of the lane's pixel for 4 channels of block kb: nine
// float4 taps straight from L1/L2 (requested one k-step ahead), bias + 9 fma in the tap order of
// yl_conv_dwh_kernel, activation -- no halo patch in LDS.  Each n-group recomputes the depthwise part (36 fma per
// 28-32 MFMAs).  Same k order and epilogues as the kernels it replaces: bit-identical.
// GW = n-groups held by ONE wave (accumulators GW x NT x 4 VGPRs): 1 = every (n-group, m-tile) pair is its own item
// and the depthwise part is recomputed per group; GW = all groups = no recomputation, 2 waves per SIMD.
template <int NT, int GW, int NW>
__global__ __launch_bounds__(NW * 64, GW == 1 ? 3 : 2) void yl_conv_dwk_kernel(YlConvP p) {
  // k-steps per weight chunk (= per barrier): 3 when a wave holds one n-group; with all groups in one wave 2 where two
  // buffers of 2 x GW x NT KiB still leave two workgroups per CU (edge_m's 244 channels: 0.359 -> 0.341 ms), else 1
  constexpr int S = GW == 1 ? 3 : (GW * NT <= 16 ? 2 : 1);
  constexpr int PCS = S * GW * NT;                           // 1 KiB pieces per chunk
  extern __shared__ __attribute__((aligned(16))) float yl_clds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  const int KB = p.KB, NTtot = p.NTtot;
  const int Cin = p.Cin, H = p.H, W = p.W, DS = p.dw_stride, pad_t = p.dw_pad_t, pad_l = p.dw_pad_l;
  const int ohw = p.OH * p.OW, OW = p.OW, M = p.M;
  const yl_act_t* const xin = p.x;
  // tap loads through a raw buffer descriptor (round 6, see yl_conv_dws_kernel): a 32-bit byte offset per tap fixed for the item +
  // a scalar k-block offset (the loop carried a select, a conditional add and a 64-bit add per tap); taps outside the image carry
  // an out-of-range offset (zeros from the range check); the channel tail is not masked (its 1x1 weights are zeros)
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<yl_act_t*>(xin), 0, (int)((long)p.B * H * W * Cin * (long)sizeof(yl_act_t)), 0x00020000);
  f32x4* wl = reinterpret_cast<f32x4*>(yl_clds);            // [2][S][GW][NT][64] float4
  float* dwl = yl_clds + (size_t)2 * PCS * 256;              // [9][Cin] taps, [Cin] bias
  const f32x4* wg = reinterpret_cast<const f32x4*>(p.wp);
  const int NC = (KB + S - 1) / S;                           // chunks per item
  const int G = NTtot / (NT * GW);                           // item groups (1 when the wave holds every n-group)
  const int bx = blockIdx.x, gx = gridDim.x;                 // gx % 8 == 0
  const int per = gx >> 3, slot = bx >> 3;
  const int tpx = (p.ntiles + 7) >> 3;
  const int band0 = (bx & 7) * tpx;
  const int band1 = (band0 + tpx) < p.ntiles ? (band0 + tpx) : p.ntiles;
  const int bt = band1 > band0 ? band1 - band0 : 0;
  const int nitems = bt * G;
  const int nmine = slot < nitems ? (nitems - 1 - slot) / per + 1 : 0;
  const long total_chunks = (long)nmine * NC;
  auto load_chunk = [&](int g, int c, int buf) {
    for (int i = wave; i < PCS; i += NW) {
      const int j = i / (GW * NT), nt = i - j * (GW * NT);
      const int kb = c * S + j;
      if (kb < KB) yl_glds16(wg + ((size_t)kb * NTtot + g * (GW * NT) + nt) * 64 + lane, wl + ((size_t)buf * PCS + i) * 64);
    }
  };
  if (total_chunks > 0) load_chunk(slot / bt, 0, 0);
  {
    const int nw = 9 * Cin;
    yl_glds_floats(p.dw_w, dwl, nw, tid, NW * 64);
    if (p.dw_b) yl_glds_floats(p.dw_b, dwl + nw, Cin, tid, NW * 64);
    else for (int i = tid; i < Cin; i += NW * 64) dwl[nw + i] = 0.0f;
  }
  long gchunk = 0;
  __syncthreads();
  const bool pre_add = (p.res || p.up) && p.act == YL_ACT_NONE;
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float dlo = (p.dw_act == YL_ACT_RELU || p.dw_act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float dhi = (p.dw_act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const int dw_act = p.dw_act;

  for (int wi = 0; wi < nmine; ++wi) {
    const int item = slot + wi * per;
    const int g = item / bt;
    const int nt0 = g * (GW * NT);
    const int tile = band0 + item - g * bt;
    YlPix px[1];
    {
      size_t lin = ((size_t)tile * NW + wave) * 16 + pl;
      px[0].valid = lin < (size_t)M;
      if (!px[0].valid) lin = (size_t)M - 1;
      px[0].lin = lin;
      const int b = (int)(lin / ohw);
      const int rem = (int)(lin - (size_t)b * ohw);
      px[0].b = b;
      px[0].oy = rem / OW;
      px[0].ox = rem - px[0].oy * OW;
    }
    f32x4 acc[GW][1][NT];
#pragma unroll
    for (int gw = 0; gw < GW; ++gw)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[gw][0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pre_add) {
      const size_t obase = px[0].lin * p.N;
      size_t up_off = 0;
      if (p.up) {
        const int uy = (px[0].oy * p.UH) / p.OH, ux = (px[0].ox * p.UW) / p.OW;
        up_off = (((size_t)px[0].b * p.UH + uy) * p.UW + ux) * p.N;
      }
#pragma unroll
      for (int gw = 0; gw < GW; ++gw)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int n = (nt0 + gw * NT + nt) * 16 + 4 * kq;
          if (n < p.N) {
            if (p.res) acc[gw][0][nt] = yl_ld4(p.res + obase + n);
            if (p.up) acc[gw][0][nt] += yl_ld4(p.up + up_off + n);
          }
        }
    }
    // nine tap pointers of the lane's pixel (the zero buffer where a tap falls outside the image, marked in `inb`)
    unsigned tp[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int iy = px[0].oy * DS - pad_t + tap / 3, ix = px[0].ox * DS - pad_l + tap % 3;
      const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
      tp[tap] = in ? (unsigned)((((px[0].b * H + iy) * W + ix) * Cin + 4 * kq) * (int)sizeof(yl_act_t)) : 0x80000000u;
    }
    auto fetch = [&](f32x4 (&dst)[9], int kb) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
#if defined_YL_F16S
        dst[tap] = __builtin_convertvector(__builtin_bit_cast(yl_h16x4, __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)tp[tap], kb * 32, 0)), f32x4);
#else
        dst[tap] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)tp[tap], kb * 64, 0));
#endif
      }
    };
    f32x4 xt[9];
    fetch(xt, 0);
    for (int c = 0; c < NC; ++c, ++gchunk) {
      const int buf = (int)(gchunk & 1);
      if (gchunk + 1 < total_chunks) {
        if (c + 1 < NC) load_chunk(g, c + 1, buf ^ 1);
        else load_chunk((item + per) / bt, 0, buf ^ 1);
      }
      const f32x4* wb = wl + (size_t)buf * PCS * 64 + lane;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        const int kb = c * S + j;
        if (kb < KB) {
          // depthwise result of block kb (tap order of yl_conv_dwh_kernel), then the next block's taps are requested
          const int cc = kb * 16 + 4 * kq;
          const int cs = cc < Cin ? cc : Cin - 4;
          const float* tapw = dwl + cs;
          f32x4 xq[1];
          xq[0] = yl_ld4(tapw + 9 * Cin);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const f32x4 w = yl_ld4(tapw + tap * Cin);
            xq[0].x = fmaf(xt[tap].x, w.x, xq[0].x); xq[0].y = fmaf(xt[tap].y, w.y, xq[0].y);
            xq[0].z = fmaf(xt[tap].z, w.z, xq[0].z); xq[0].w = fmaf(xt[tap].w, w.w, xq[0].w);
          }
          xq[0] = yl_actc(xq[0], dw_act, dlo, dhi);
          if (kb + 1 < KB) fetch(xt, kb + 1);
#pragma unroll
          for (int gw = 0; gw < GW; ++gw) {
            f32x4 wq[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wq[nt] = wb[((j * GW + gw) * NT + nt) * 64];
            yl_mma_step<NT, 1>(wq, xq, acc[gw]);
          }
        }
      }
      __syncthreads();             // every wave is done with `buf`; the copies into the other buffer have landed
    }
#pragma unroll
    for (int gw = 0; gw < GW; ++gw) {
      if (!pre_add && (p.res || p.up || YL_SMOOTH(p.act))) yl_epi_generic<NT, 1>(p, acc[gw], px, nt0 + gw * NT, kq);
      else yl_epi_fast<NT, 1>(p, acc[gw], px, nt0 + gw * NT, kq, lo, hi, true);
    }
  }
}

template <int NT, int GW, int NW>
static hipError_t dwk_go(const YlConvP& p0, hipStream_t st, bool attr_only) {
  if (attr_only)
    return hipFuncSetAttribute((const void*)yl_conv_dwk_kernel<NT, GW, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  YlConvP p = p0;
  p.ntiles = (int)(((long)p.M + 16 * NW - 1) / (16 * NW));
  const size_t lds = (size_t)2 * (GW == 1 ? 3 : (GW * NT <= 16 ? 2 : 1)) * GW * NT * 1024 + (((size_t)10 * p.Cin + 3) & ~(size_t)3) * 4;
  if (lds > 96 * 1024) return hipErrorNotSupported;
  const int res = yl_resident_blocks_n(yl_conv_dwk_kernel<NT, GW, NW>, NW * 64, lds);
  const int G = p.NTtot / (NT * GW);
  int gx = res & ~7;
  while (gx > 8 && gx - 8 >= p.ntiles * G) gx -= 8;
  hipLaunchKernelGGL((yl_conv_dwk_kernel<NT, GW, NW>), dim3(gx), dim3(NW * 64), lds, st, p);
  return hipGetLastError();
}

template <int NTT>
static hipError_t dwl_go(const YlConvP& p0, hipStream_t st, bool attr_only);

// depthwise 3x3 (stride 1 / 2) -> 1x1 with K >= 192, N % 4 == 0 and an n-tile count that is a multiple of 7 or 8,
// single problem.  hipErrorNotSupported otherwise (yl_conv_mfma_kernel's streamed mode then runs the layer).
hipError_t yl_launch_conv_dwk(const YlConvP& p, hipStream_t st) {
  if (p.dw_k != 3 || (p.N & 3) || p.dec_boxes || p.C1 > 0 || p.KB < 12 || p.NTtot <= 8) return hipErrorNotSupported;
  if ((size_t)p.B * p.H * p.W * p.Cin * sizeof(yl_act_t) >= ((size_t)1 << 31)) return hipErrorNotSupported;   // 32-bit byte offsets
  // developer A/B ("dev_select" bits 5-6): 2 = off, 1 = one n-group per item (depthwise recomputed per group),
  // 0 = default: a wave holds every n-group
  const int sel = YL_DEV_DWK(p.dev) == 2 ? 0 : (YL_DEV_DWK(p.dev) == 1 ? 1 : 2);
  if (sel == 0) return hipErrorNotSupported;
  // window-in-LDS form (yl_conv_dwl_kernel): stride 1, pad 1, grids that fill the 8 x 8-pixel windows to >= 80 % (20 x 20: 69 %, slower than the tap-load kernel), >= 1.5 items
  // per CU (40 x 40 at B = 32: 0.116 -> 0.098 ms); "dev_select" bit 14 = off, bit 15 = on every grid (the bitwise test)
  // (the fp16-storage unit keeps the tap-load kernel: the windows are raw LDS-DMA copies of the tensor's bytes)
  if (!(defined_YL_F16S) && !(p.dev & YL_DEV_DWL_OFF) && p.dw_stride == 1 && p.dw_pad_t == 1 && p.dw_pad_l == 1 && !p.scale &&
      (size_t)p.B * p.H * p.W * p.Cin < ((size_t)1 << 31)) {
    const long wins = (long)p.B * ((p.OW + 7) >> 3) * ((p.OH + 7) >> 3);
    if (((long)p.B * p.OH * p.OW * 10 >= wins * 64 * 8 && wins >= 3 * YL_NUM_CU) || (p.dev & YL_DEV_DWL_ALL)) {
      if (p.NTtot == 16) return dwl_go<16>(p, st, false);
      if (p.NTtot == 21) return dwl_go<21>(p, st, false);
    }
  }
  if (p.NTtot == 21) return sel == 2 ? dwk_go<7, 3, 4>(p, st, false) : dwk_go<7, 1, 4>(p, st, false);
  if (p.NTtot == 16) return sel == 2 ? dwk_go<8, 2, 4>(p, st, false) : dwk_go<8, 1, 4>(p, st, false);
  if (p.NTtot % 7 == 0) return dwk_go<7, 1, 4>(p, st, false);
  if (p.NTtot % 8 == 0) return dwk_go<8, 1, 4>(p, st, false);
  return hipErrorNotSupported;
}

// ------------------------------------------------------------------------------------------------
// Depthwise 3x3 (stride 1, pad 1) -> wide 1x1 with the INPUT WINDOW IN LDS (round 5): the layers of yl_conv_dwk_kernel
// on grids that fill 8 x 8-pixel windows.  yl_conv_dwk_kernel fetches nine float4 taps per lane and k-block straight
// from L1/L2 with the MFMA lane layout (lane = channel group * 16 + pixel): the four lanes of a quad sit on four
// different pixels, i.e. four cache lines, so a tap load keeps the texture addresser busy for 64 cycles and the nine of
// them for more cycles than the 64 MFMAs they feed (244 -> 244 @80x80: 55 TFLOP/s).  Here, with the machinery of
// yl_conv_wino2_kernel:
//   window   a workgroup item is TWO 8 x 8-pixel output windows (8 waves x one 4 x 4-pixel MFMA m-tile); their 10 x 10-pixel
//            input windows (64 B per pixel and k-block) land in LDS by asynchronous LDS-DMA copies whose quads read the 64
//            contiguous bytes of ONE pixel (2 copy instructions per wave and k-block; every input pixel once per item instead
//            of once per tap that touches it), double-buffered, requested a k-block ahead;
//   B        the lane's nine taps by ds_read_b128 from the window (slot = 4 P + (kq ^ 2 ((P >> 2) & 1)), P = y * 12 + x: a
//            lane group holds pixel rows {0,3} with one channel group and {1,2} with its neighbour -- conflict-free, see
//            yl_conv_wino2_kernel), tap weights + bias from LDS, the fmaf chain and activation of yl_conv_dwk_kernel;
//   A        the 1x1 weights of a k-block (all n-tiles: a wave accumulates every output channel of its 16 pixels, as in
//            yl_conv_dwk_kernel's GW = all form) shared by the 8 waves through LDS, TRIPLE-buffered;
//   barrier  ONE per k-block, in the middle of its MFMAs: behind it window(kb + 1) and weights(kb + 1) are complete, the
//            wave requests window(kb + 2) / weights(kb + 2) into the buffers of blocks kb / kb - 1, builds B(kb + 1) and goes
//            on with the second half of block kb's MFMAs -- the MFMA stream runs on across k-blocks.
// Same tap order, k order and epilogues as yl_conv_dwk_kernel: BIT-IDENTICAL to it ("dev_select" bit 14 keeps the old kernel).
template <int NTT>
__global__ __launch_bounds__(512, 2) void yl_conv_dwl_kernel(YlConvP p) {
  constexpr int RP = 12, RM = 512;
  constexpr int NTP = (NTT + 7) & ~7;                            // weight pieces per buffer (8 waves x NTP / 8 copies)
  // MFMA groups of <= 4 n-tiles, an even number of them (their A fragments alternate between two register sets across
  // k-blocks): 16 = 4 x 4, 21 = 3 x 4 + 3 x 3
  constexpr int NCH = NTT == 16 ? 4 : 6;
  static_assert(NTT == 16 || NTT == 21, "n-tile groups");
  auto c0of = [](int c) { return NTT == 16 ? 4 * c : (c <= 3 ? 4 * c : 12 + 3 * (c - 3)); };
  extern __shared__ __attribute__((aligned(16))) float yl_clds[];
  f32x4* const Wl = reinterpret_cast<f32x4*>(yl_clds);           // [3][NTP][64]
  f32x4* const Rl = Wl + 3 * NTP * 64;                           // [2 buffers][2 windows][RM]
  float* const dwl = reinterpret_cast<float*>(Rl + 2 * 2 * RM);  // [KB][10][16]: taps 0..8 + bias of a k-block's channels (zeros beyond Cin)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  const int KB = p.KB, NTtot = p.NTtot, Cin = p.Cin, H = p.H, W = p.W, OH = p.OH, OW = p.OW;
  const int WX = (OW + 7) >> 3, WY = (OH + 7) >> 3;
  const int wimg = WX * WY;
  const long WTOT = (long)p.B * wimg;                            // windows
  const yl_act_t* const xin = p.x;
  const int wgmax = KB * NTtot - 1;                              // last weight piece
  const int bx = blockIdx.x, gx = gridDim.x;                     // gx % 8 == 0
  const int per = gx >> 3, slot = bx >> 3;
  const int tpx = (p.ntiles + 7) >> 3;                           // items per XCD band
  const int band0 = (bx & 7) * tpx;
  const int band1 = (band0 + tpx) < p.ntiles ? (band0 + tpx) : p.ntiles;
  // copy role of the lane: slot rs = wave * 64 + lane of both windows
  const int rs = wave * 64 + lane;
  const int rP = rs >> 2, rkq = (rs & 3) ^ (((rP >> 2) & 1) << 1);
  const int ry = rP / RP, rx = rP - ry * RP;
  const bool rpix = ry < 10 && rx < 10;
  // compute role: window wsel, 4 x 4-pixel block (qy, qx) of it, pixel (sy, sx) of the block
  const int wsel = wave >> 2, qy = (wave >> 1) & 1, qx = wave & 1;
  const int sy = pl >> 2, sx = pl & 3;
  int ts[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int P = (4 * qy + sy + tap / 3) * RP + 4 * qx + sx + tap % 3;
    ts[tap] = wsel * RM + 4 * P + (kq ^ (((P >> 2) & 1) << 1));
  }
  // the tap image k-block-major, so that a lane's ten tap / bias reads of a block are ONE address + immediates
  for (int i = tid; i < KB * 160; i += 512) {
    const int kb = i / 160, r = i - kb * 160, t = r >> 4, ch = kb * 16 + (r & 15);
    dwl[i] = ch < Cin ? (t < 9 ? p.dw_w[(size_t)t * Cin + ch] : (p.dw_b ? p.dw_b[ch] : 0.0f)) : 0.0f;
  }
  // Operand streams through raw buffer descriptors (as yl_conv_wino2_kernel, round 6): the lane's part of an address is a 32-bit
  // byte offset fixed for the item (window) or the launch (weights), the k-block part is scalar; lanes outside the image carry an
  // out-of-range offset and the copy writes zeros.  The channel tail of the last k-block is not masked: those lanes copy the next
  // pixel's first channels (the arenas end in 256 spare bytes), their tap weights here and their 1x1 weights are zeros.
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<yl_act_t*>(xin), 0, (int)((long)p.B * H * W * Cin * (long)sizeof(yl_act_t)), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.wp), 0, (int)((long)KB * NTtot * 1024), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int lane16 = lane * 16;
  const int sh = KB & 1;                                         // window(kb) lives in buffer (kb + KB) & 1: see yl_conv_wino2_kernel
  const bool pre_add = (p.res || p.up) && p.act == YL_ACT_NONE;
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float dlo = (p.dw_act == YL_ACT_RELU || p.dw_act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float dhi = (p.dw_act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const int dw_act = p.dw_act;

  unsigned voff[2];
  auto issue_win = [&](int kb, int buf) {
#pragma unroll
    for (int w2 = 0; w2 < 2; ++w2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(Rl + (buf * 2 + w2) * RM + wave * 64), 16,
                                               (int)voff[w2], kb * 16 * (int)sizeof(yl_act_t), 0, 0);
  };
  auto issue_wts = [&](int kb, int wb) {
#pragma unroll
    for (int j = 0; j < NTP / 8; ++j) {
      const int i = wave + 8 * j;                                // (pieces NTT..NTP-1: pad, any valid source)
      int src = kb * NTtot + (i < NTT ? i : 0);
      src = src < wgmax ? src : wgmax;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(Wl + ((size_t)wb * NTP + i) * 64), 16,
                                               lane16, src * 1024, 0, 0);
    }
  };
  auto make_b = [&](int kb, int buf) {
    const float* tapw = dwl + kb * 160 + 4 * kq;
    const f32x4* const wb = Rl + buf * 2 * RM;
    f32x4 xq = yl_ld4(tapw + 9 * 16);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const f32x4 x = wb[ts[tap]];
      const f32x4 w = yl_ld4(tapw + tap * 16);
      xq.x = fmaf(x.x, w.x, xq.x); xq.y = fmaf(x.y, w.y, xq.y);
      xq.z = fmaf(x.z, w.z, xq.z); xq.w = fmaf(x.w, w.w, xq.w);
    }
    return yl_actc(xq, dw_act, dlo, dhi);
  };

  for (int item = band0 + slot, wi = 0; item < band1; item += per, ++wi) {   // (wi: the stamp builds' item counter)
    // the two windows of the item, the lane's output pixel, the lane's copy sources
    YlPix px[1];
    {
      const long wi = (long)item * 2 + wsel;
      const bool wv = wi < WTOT;
      const long wc = wv ? wi : WTOT - 1;
      const int b = (int)(wc / wimg);
      const int r = (int)(wc - (long)b * wimg);
      const int wy = r / WX, wx = r - wy * WX;
      int oy = 8 * wy + 4 * qy + sy, ox = 8 * wx + 4 * qx + sx;
      px[0].valid = wv && oy < OH && ox < OW;
      if (oy >= OH) oy = OH - 1;
      if (ox >= OW) ox = OW - 1;
      px[0].b = b; px[0].oy = oy; px[0].ox = ox;
      px[0].lin = ((size_t)b * OH + oy) * OW + ox;
    }
#pragma unroll
    for (int w2 = 0; w2 < 2; ++w2) {
      const long wi = (long)item * 2 + w2;
      const int b = (int)(wi / wimg);
      const int r = (int)(wi - (long)b * wimg);
      const int wy = r / WX, wx = r - wy * WX;
      const int gy = 8 * wy - 1 + ry, gxx = 8 * wx - 1 + rx;
      const bool in = rpix && wi < WTOT && gy >= 0 && gy < H && gxx >= 0 && gxx < W;
      voff[w2] = in ? (unsigned)((((b * H + gy) * W + gxx) * Cin + 4 * rkq) * (int)sizeof(yl_act_t)) : OOB;
    }
    f32x4 acc[1][NTT];
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) acc[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pre_add) {
      const size_t obase = px[0].lin * p.N;
      size_t up_off = 0;
      if (p.up) {
        const int uy = (px[0].oy * p.UH) / p.OH, ux = (px[0].ox * p.UW) / p.OW;
        up_off = (((size_t)px[0].b * p.UH + uy) * p.UW + ux) * p.N;
      }
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        const int n = nt * 16 + 4 * kq;
        if (n < p.N) {
          if (p.res) acc[0][nt] = yl_ld4(p.res + obase + n);
          if (p.up) acc[0][nt] += yl_ld4(p.up + up_off + n);
        }
      }
    }
    __syncthreads();                                              // the previous item's last weight reads are done
    issue_win(0, sh); issue_wts(0, 0);
    issue_win(1, sh ^ 1); issue_wts(1, 1);                        // (KB >= 12: the launcher)
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0)
    __syncthreads();
    f32x4 xq[1];
    xq[0] = make_b(0, sh);
    int wb = 0;
    f32x4 wq[2][4];                                               // A fragments of the current / the next MFMA group
    auto read_a = [&](int wbuf, int c, f32x4 (&dst)[4]) {
      const f32x4* const wl = Wl + (size_t)wbuf * NTP * 64 + lane;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (c0of(c) + i < c0of(c + 1)) dst[i] = wl[(c0of(c) + i) * 64];
    };
    read_a(0, 0, wq[0]);
    // One k-block.  Top: the barrier behind which window(kb + 1) / weights(kb + 1) are complete (requested a k-block ago)
    // and nobody reads window(kb) / weights(kb - 1) any more -- their buffers take the requests of block kb + 2.  Then the
    // MFMA groups; in front of group c the wave issues the LDS reads of what it needs one group later -- the A fragments of
    // group c + 1 (of block kb + 1's group 0 at the end: complete since this block's barrier) and three taps + tap weights
    // of B(kb + 1) -- and behind the group's MFMAs their arithmetic: no LDS round trip is waited for, the tap fma chain sits
    // between MFMA groups (a first version built B(kb + 1) in one piece between two halves of the block's MFMAs: both waves
    // of a SIMD waited there for 19 LDS reads with nothing to issue).  MODE 0 any block, 1 the second-to-last (no
    // requests), 2 the last (no next B): branch-free bodies.
    auto kblock = [&](int kb, auto par, auto mode) {
      constexpr int PAR = decltype(par)::value;                   // buffer of window(kb) (compile-time: LDS offsets are immediates)
      constexpr int MODE = decltype(mode)::value;
      const int wb1 = wb == 2 ? 0 : wb + 1;
      // the wave's own copies first: the compiler's wait-count pass does not carry the LDS-DMA requests of the previous
      // iteration across the loop's back edge (the barrier here came out with lgkmcnt(0) only, and two runs of edge_m at
      // B = 32 differed in a few bits)
      WINO_STAMP(kb * 7 + 0);
      __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
      __syncthreads();
      WINO_STAMP(kb * 7 + 1);
      if (MODE == 0) {
        issue_win(kb + 2, PAR);
        issue_wts(kb + 2, wb >= 1 ? wb - 1 : 2);                  // (wb + 2) % 3
      }
      WINO_STAMP(kb * 7 + 2);
      const float* const tapw = dwl + (kb + 1) * 160 + 4 * kq;
      const f32x4* const win = Rl + (PAR ^ 1) * 2 * RM;
      f32x4 xn = xq[0];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        f32x4 tx[3], tw[3];
        if (c + 1 < NCH) read_a(wb, c + 1, wq[(c + 1) & 1]);
        else if (MODE < 2) read_a(wb1, 0, wq[0]);
        if (MODE < 2 && c < 3) {
          if (c == 0) xn = yl_ld4(tapw + 9 * 16);
#pragma unroll
          for (int t = 0; t < 3; ++t) { tx[t] = win[ts[3 * c + t]]; tw[t] = yl_ld4(tapw + (3 * c + t) * 16); }
        }
        __builtin_amdgcn_sched_barrier(0);
#if YL_BF16
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (c0of(c) + i < c0of(c + 1)) {
            const f32x4 w1[1] = {wq[c & 1][i]};
            yl_mma_step<1, 1>(w1, xq, *reinterpret_cast<f32x4 (*)[1][1]>(&acc[0][c0of(c) + i]));
          }
#else
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (c0of(c) + i < c0of(c + 1))
              acc[0][c0of(c) + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[c & 1][i][st], xq[0][st], acc[0][c0of(c) + i], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);                        // (else the fma chain moves up to the group's first MFMA and waits there
        if (MODE < 2 && c < 3) {                                  //  for the reads issued a moment before)
#pragma unroll
#if defined(YL_DWL_PK) && !YL_DWL_PK                                 // (A/B builds)
          for (int t = 0; t < 3; ++t) xn = __builtin_elementwise_fma(tx[t], tw[t], xn);
#else
          for (int t = 0; t < 3; ++t) xn = yl_pk_fma4(tx[t], tw[t], xn);     // two v_pk_fma_f32 by name: the same fma per component
#endif
          if (c == 2) xn = yl_actc(xn, dw_act, dlo, dhi);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c < 4) WINO_STAMP(kb * 7 + 3 + c);
      }
      xq[0] = xn;
      wb = wb1;
    };
    {
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      int kb = 0;
      if (sh) { kblock(0, I1{}, I0{}); kb = 1; }                  // odd KB: block 0 on its own, out of buffer 1
      for (; kb + 3 < KB; kb += 2) {
        kblock(kb, I0{}, I0{});
        kblock(kb + 1, I1{}, I0{});
      }
      kblock(KB - 2, I0{}, I1{});
      kblock(KB - 1, I1{}, I2{});
    }
    if (!pre_add && (p.res || p.up || YL_SMOOTH(p.act))) yl_epi_generic<NTT, 1>(p, acc, px, 0, kq);
    else yl_epi_fast<NTT, 1>(p, acc, px, 0, kq, lo, hi, true);
  }
}

template <int NTT>
static hipError_t dwl_go(const YlConvP& p0, hipStream_t st, bool attr_only) {
  constexpr int NTP = (NTT + 7) & ~7;
  if (attr_only)
    return hipFuncSetAttribute((const void*)yl_conv_dwl_kernel<NTT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  YlConvP p = p0;
  const long WTOT = (long)p.B * ((p.OW + 7) >> 3) * ((p.OH + 7) >> 3);
  p.ntiles = (int)((WTOT + 1) / 2);
  const size_t lds = ((size_t)3 * NTP * 64 + 4 * 512) * 16 + (size_t)p.KB * 160 * 4;
  if (lds > 128 * 1024) return hipErrorNotSupported;
  if ((size_t)p.B * p.H * p.W * p.Cin * sizeof(yl_act_t) >= ((size_t)1 << 31)) return hipErrorNotSupported;   // 32-bit byte offsets
  const int res = yl_resident_blocks_n(yl_conv_dwl_kernel<NTT>, 512, lds);
  int gx = res & ~7;
  if (gx < 8) gx = 8;
  while (gx > 8 && gx - 8 >= p.ntiles) gx -= 8;
  hipLaunchKernelGGL((yl_conv_dwl_kernel<NTT>), dim3(gx), dim3(512), lds, st, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Depthwise DK x DK (stride DS) -> 1x1 with MANY channels on the depthwise side (round 5): the `conv_dw` -> `conv_pwl` halves of
// timm's EfficientNet-Lite inverted-residual blocks behind model_v2.py:94-100 (yololite_m = tf_efficientnet_lite2 stages 4-6:
// 528 / 720 / 1248 channels, 5x5 stride 1 and 2, 3x3 at 1248 -> 352).  Neither fused kernel
// covered them -- yl_conv_dwt_kernel keeps the taps of ALL channels in LDS (26 x 1248 floats = 130 KB) and reads its 1x1
// weights per wave from L1/L2 (<= 6 n-tiles), yl_conv_dwk_kernel fetches nine taps per lane from L1/L2 (25 at 5x5: bound by
// the vector-memory path) -- so the depthwise conv ran as its own launch (yl_dw_tile_kernel: ten launches, ~0.8 ms of the
// 12 ms step at B = 32, the expanded tensor written and re-read).  This kernel joins the two recipes:
//   from yl_conv_dwk_kernel  the 1x1 weight stream double-buffered through LDS in chunks of S k-steps shared by the NW waves
//                            of a workgroup (asynchronous copies, one barrier per chunk), a wave holds EVERY n-tile of its
//                            pixels (GW x NT accumulator sets), items dealt in XCD bands;
//   from yl_conv_dwt_kernel  wave = one 4x4-pixel tile, the (3 DS + DK)^2 halo patch of a 16-channel block staged through the
//                            wave's private LDS region (4 / 8 / 3 coalesced float4 loads per lane, requested one k-step
//                            ahead), B = act(bias + sum_taps w * x) from conflict-free ds_read_b128;
//   new                      the tap weights + depthwise bias of a k-step (26 x 16 floats) ride in the weight chunk: LDS holds
//                            2 x S k-steps of them, whatever the channel count.
// Same tap order (dy, dx), k order and epilogues as yl_dw_tile_kernel + yl_conv_pws_kernel: the depthwise value of a channel is
// the same fmaf chain, the 1x1 sums its k blocks in ascending order -> bit-identical to the two launches it replaces.
template <int NT, int GW, int DK, int DS, int NW>
__global__ __launch_bounds__(NW * 64, 2) void yl_conv_dws_kernel(YlConvP p) {
  constexpr int NTW = NT * GW;                                        // n-tiles a wave accumulates (all of the layer's)
  constexpr int S = NTW <= 8 ? 2 : 1;                                // k-steps per chunk (= per barrier)
  constexpr int TAPS = DK * DK, TQ = (TAPS + 1) * 4;                 // float4s of taps + bias per k-step
  constexpr int TQP = ((TQ + 63) / 64) * 64;                         // ... padded to whole 64-lane copies
  constexpr int HP = 3 * DS + DK;                                    // halo patch rows = columns
  constexpr int PITCHF = ((HP * 16 + 7) / 64) * 64 + 56;             // row pitch in floats (see yl_conv_dwh_kernel)
  constexpr int HF4 = HP * HP * 4, NSLOT = (HF4 + 63) / 64;
  constexpr int PCS = S * NTW + S * (TQP / 64);                      // 1 KiB pieces per chunk: weights, then taps
  extern __shared__ __attribute__((aligned(16))) float yl_clds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  const int KB = p.KB, NTtot = p.NTtot, Cin = p.Cin, H = p.H, W = p.W, OH = p.OH, OW = p.OW;
  const yl_act_t* const xin = p.x;
  // staging loads through a raw buffer descriptor (round 6, as yl_conv_wino2_kernel): a 32-bit byte offset per slot fixed for the
  // item + a scalar k-block offset -- the loop carried 8 64-bit adds and 16 selects per k-block for its four loads; slots outside
  // the image carry an out-of-range offset (the load returns zeros); the channel tail is not masked (tap weights and 1x1 weights
  // of those channels are zeros, the arenas end in 256 spare bytes)
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<yl_act_t*>(xin), 0, (int)((long)p.B * H * W * Cin * (long)sizeof(yl_act_t)), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  f32x4* wl = reinterpret_cast<f32x4*>(yl_clds);                     // [2][S][NTW][64] float4
  f32x4* tl = wl + (size_t)2 * S * NTW * 64;                         // [2][S][TQP] float4: row t = tap t (t = TAPS: bias), 4 quads
  float* halo = reinterpret_cast<float*>(tl + (size_t)2 * S * TQP) + wave * (HP * PITCHF);
  const f32x4* wg = reinterpret_cast<const f32x4*>(p.wp);
  const int NC = (KB + S - 1) / S;
  const int twn = OW >> 2, tiles_img = twn * (OH >> 2);
  const long ntiles4 = (long)p.B * tiles_img;                        // 4x4 tiles; p.ntiles = items of NW tiles
  const int bx = blockIdx.x, gx = gridDim.x;                         // gx % 8 == 0
  const int per = gx >> 3, slot = bx >> 3;
  const int tpx = (p.ntiles + 7) >> 3;
  const int band0 = (bx & 7) * tpx;
  const int band1 = (band0 + tpx) < p.ntiles ? (band0 + tpx) : p.ntiles;
  const int bt = band1 > band0 ? band1 - band0 : 0;
  const int nmine = slot < bt ? (bt - 1 - slot) / per + 1 : 0;
  const long total_chunks = (long)nmine * NC;
  auto load_chunk = [&](int c, int buf) {
    for (int i = wave; i < PCS; i += NW) {
      if (i < S * NTW) {
        const int j = i / NTW, nt = i - j * NTW;
        const int kb = c * S + j;
        if (kb < KB) yl_glds16(wg + ((size_t)kb * NTtot + (nt < NTtot ? nt : NTtot - 1)) * 64 + lane, wl + ((size_t)buf * S * NTW + i) * 64);
      } else {
        const int i2 = i - S * NTW;
        const int j = i2 / (TQP / 64), r = i2 - j * (TQP / 64);
        const int kb = c * S + j;
        const int idx = r * 64 + lane, t = idx >> 2, ch = kb * 16 + (idx & 3) * 4;
        if (kb < KB && idx < TQ) {
          const float* src = (ch < Cin) ? (t < TAPS ? p.dw_w + (size_t)t * Cin + ch : (p.dw_b ? p.dw_b + ch : reinterpret_cast<const float*>(p.zeros))) : reinterpret_cast<const float*>(p.zeros);
          yl_glds16(src, tl + ((size_t)(buf * S + j) * TQP + r * 64));
        }
      }
    }
  };
  if (total_chunks > 0) load_chunk(0, 0);
  long gchunk = 0;
  // staging slots of this lane: halo pixel / channel quad -> LDS offset
  int s_lo[NSLOT];
  bool s_ok[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int e = j * 64 + lane;
    s_ok[j] = e < HF4;
    const int hp = (s_ok[j] ? e : 0) >> 2, quad = e & 3;
    const int hr = hp / HP, hc = hp - hr * HP;
    s_lo[j] = hr * PITCHF + hc * 16 + quad * 4;
  }
  __syncthreads();
  const bool pre_add = (p.res || p.up) && p.act == YL_ACT_NONE;
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float dlo = (p.dw_act == YL_ACT_RELU || p.dw_act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float dhi = (p.dw_act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const int dw_act = p.dw_act;
  const int rbase = ((pl >> 2) * DS) * PITCHF + ((pl & 3) * DS) * 16 + 4 * kq;

  for (int wi = 0; wi < nmine; ++wi) {
    const int item = band0 + slot + wi * per;
    long t4 = (long)item * NW + wave;
    const bool tvalid = t4 < ntiles4;
    if (!tvalid) t4 = ntiles4 - 1;
    const int b = (int)(t4 / tiles_img);
    const int trem = (int)(t4 - (long)b * tiles_img);
    const int tyi = trem / twn, txi = trem - tyi * twn;
    YlPix px[1];
    px[0].b = b; px[0].oy = 4 * tyi + (pl >> 2); px[0].ox = 4 * txi + (pl & 3); px[0].valid = tvalid;
    px[0].lin = ((size_t)b * OH + px[0].oy) * OW + px[0].ox;
    f32x4 acc[GW][1][NT];
#pragma unroll
    for (int gw = 0; gw < GW; ++gw)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[gw][0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pre_add) {
      const size_t obase = px[0].lin * p.N;
      size_t up_off = 0;
      if (p.up) {
        const int uy = (px[0].oy * p.UH) / p.OH, ux = (px[0].ox * p.UW) / p.OW;
        up_off = (((size_t)px[0].b * p.UH + uy) * p.UW + ux) * p.N;
      }
#pragma unroll
      for (int gw = 0; gw < GW; ++gw)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int n = (gw * NT + nt) * 16 + 4 * kq;
          if (n < p.N) {
            if (p.res) acc[gw][0][nt] = yl_ld4(p.res + obase + n);
            if (p.up) acc[gw][0][nt] += yl_ld4(p.up + up_off + n);
          }
        }
    }
    unsigned goff[NSLOT];
    {
      const int iy0 = 4 * tyi * DS - p.dw_pad_t, ix0 = 4 * txi * DS - p.dw_pad_l;
#pragma unroll
      for (int j = 0; j < NSLOT; ++j) {
        const int e = j * 64 + lane;
        const int hp = (e < HF4 ? e : 0) >> 2;
        const int hr = hp / HP, hc = hp - hr * HP;
        const int iy = iy0 + hr, ix = ix0 + hc;
        const bool in = e < HF4 && iy >= 0 && iy < H && ix >= 0 && ix < W;
        goff[j] = in ? (unsigned)((((b * H + iy) * W + ix) * Cin + (lane & 3) * 4) * (int)sizeof(yl_act_t)) : OOB;
      }
    }
    auto stage_load = [&](int kb, f32x4 (&r)[NSLOT]) {
#pragma unroll
      for (int j = 0; j < NSLOT; ++j) {
#if defined_YL_F16S
        r[j] = __builtin_convertvector(__builtin_bit_cast(yl_h16x4, __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)goff[j], kb * 32, 0)), f32x4);
#else
        r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)goff[j], kb * 64, 0));
#endif
      }
    };
    auto stage_store = [&](const f32x4 (&r)[NSLOT]) {
#pragma unroll
      for (int j = 0; j < NSLOT; ++j)
        if (s_ok[j]) *reinterpret_cast<f32x4*>(halo + s_lo[j]) = r[j];
    };
    f32x4 stg[NSLOT];
    stage_load(0, stg);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // previous item's tap reads are complete
    stage_store(stg);
    for (int c = 0; c < NC; ++c, ++gchunk) {
      const int buf = (int)(gchunk & 1);
      if (gchunk + 1 < total_chunks) load_chunk(c + 1 < NC ? c + 1 : 0, buf ^ 1);
#pragma unroll
      for (int j = 0; j < S; ++j) {
        const int kb = c * S + j;
        if (kb < KB) {
          const bool more = kb + 1 < KB;
          if (more) stage_load(kb + 1, stg);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // halo writes (all lanes) -> tap reads
          const f32x4* tw = tl + (size_t)(buf * S + j) * TQP + kq;
          f32x4 xq[1];
          xq[0] = tw[TAPS * 4];
#pragma unroll 1
          for (int dy = 0; dy < DK; ++dy) {                          // one tap row at a time bounds the register footprint
#pragma unroll
            for (int dx = 0; dx < DK; ++dx) {
              // (ablation builds, results wrong: -DYL_DWS_ABL=1 one tap-weight read per row, =2 one tap read per row, =3 both, =4 no fma chain)
#if defined(YL_DWS_ABL) && (YL_DWS_ABL & 1)
              const f32x4 w = tw[(dy * DK) * 4];
#else
              const f32x4 w = tw[(dy * DK + dx) * 4];
#endif
#if defined(YL_DWS_ABL) && (YL_DWS_ABL & 2)
              const f32x4 v = *reinterpret_cast<const f32x4*>(halo + rbase + dy * PITCHF);
#else
              const f32x4 v = *reinterpret_cast<const f32x4*>(halo + rbase + dy * PITCHF + dx * 16);
#endif
#if defined(YL_DWS_ABL) && (YL_DWS_ABL & 4)
              if (dx == 0) { xq[0] += v * w; }
#else
              xq[0].x = fmaf(v.x, w.x, xq[0].x); xq[0].y = fmaf(v.y, w.y, xq[0].y);
              xq[0].z = fmaf(v.z, w.z, xq[0].z); xq[0].w = fmaf(v.w, w.w, xq[0].w);
#endif
            }
          }
          xq[0] = yl_actc(xq[0], dw_act, dlo, dhi);
          // channel tail (c >= Cin): the packed 1x1 weights of those k slots are zero (and the taps were copied from zeros)
          const f32x4* wb = wl + (size_t)(buf * S + j) * NTW * 64 + lane;
#pragma unroll
          for (int gw = 0; gw < GW; ++gw) {
            f32x4 wq[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wq[nt] = wb[(gw * NT + nt) * 64];
            yl_mma_step<NT, 1>(wq, xq, acc[gw]);
          }
          if (more) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // this block's tap reads are complete
            stage_store(stg);
          }
        }
      }
      __syncthreads();             // every wave is done with `buf`; the copies into the other buffer have landed
    }
#pragma unroll
    for (int gw = 0; gw < GW; ++gw) {
      if (!pre_add && (p.res || p.up || YL_SMOOTH(p.act))) yl_epi_generic<NT, 1>(p, acc[gw], px, gw * NT, kq);
      else yl_epi_fast<NT, 1>(p, acc[gw], px, gw * NT, kq, lo, hi, true);
    }
  }
}

static size_t yl_dws_lds(int ntw, int dk, int ds, int nw) {
  const int s = ntw <= 8 ? 2 : 1, tqp = ((dk * dk + 1) * 4 + 63) / 64 * 64, hp = 3 * ds + dk;
  const int pitch = ((hp * 16 + 7) / 64) * 64 + 56;
  return ((size_t)2 * s * ntw * 256 + (size_t)2 * s * tqp * 4 + (size_t)nw * hp * pitch) * 4;
}

template <int NT, int GW, int DK, int DS, int NW>
static hipError_t dws_go(const YlConvP& p0, hipStream_t st, bool attr_only) {
  if (attr_only)
    return hipFuncSetAttribute((const void*)yl_conv_dws_kernel<NT, GW, DK, DS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  YlConvP p = p0;
  const long t4 = (long)p.B * (p.OH >> 2) * (p.OW >> 2);
  p.ntiles = (int)((t4 + NW - 1) / NW);
  const size_t lds = yl_dws_lds(NT * GW, DK, DS, NW);
  const int res = yl_resident_blocks_n(yl_conv_dws_kernel<NT, GW, DK, DS, NW>, NW * 64, lds);
  int gx = res & ~7;
  if (gx < 8) gx = 8;                                            // (occupancy query below 8: never a grid of 0 -- ADVICE r05)
  while (gx > 8 && gx - 8 >= p.ntiles) gx -= 8;
  hipLaunchKernelGGL((yl_conv_dws_kernel<NT, GW, DK, DS, NW>), dim3(gx), dim3(NW * 64), lds, st, p);
  return hipGetLastError();
}

// instantiated: (n-tiles per accumulator set, sets, depthwise k, stride): 8 n-tiles (N <= 128), 13 (N <= 208), 2 x 11 (N <= 352).
// Measured on yololite_m B = 32 (eager, ms; two launches -> this kernel): 528 -> 120 5x5 @40x40 0.191 -> 0.160, 720 -> 120 0.246 ->
// 0.217, 720 -> 208 5x5 s2 0.131 -> 0.127, 1248 -> 208 5x5 @20x20 0.168 -> 0.164, 1248 -> 352 3x3 (22 n-tiles) 0.188 -> 0.228.
// 45-50 TFLOP/s: the 25 tap + 25 tap-weight ds_read_b128 per k-step make it LDS-bound at two waves per SIMD (S = 1, 8 waves per
// workgroup and a fully unrolled tap loop were measured: 0.220 / 0.210 (but 0.233 at 13 n-tiles) / 0.226 against 0.217).  What
// counts is the step with two batches in flight, where launches and HBM traffic are the currency: 2.58 k images/s with the
// eleven stand-alone depthwise launches, 2.65 k with ten of them fused, 2.72 k with all eleven (the 22 n-tile layer included,
// although it is slower in isolation) -- 2.9 GB of expanded-tensor traffic fewer per step.
#define YL_DWS_SHAPES(X) X(8, 1, 5, 1) X(8, 1, 5, 2) X(13, 1, 5, 1) X(13, 1, 5, 2) X(8, 1, 3, 1) X(13, 1, 3, 1) X(11, 2, 3, 1) X(11, 2, 5, 1)

#if !YL_BF16
// depthwise -> 1x1 layers this kernel takes: >= 12 k-blocks (below that yl_conv_dwt / dwh / dwk_kernel and the 32 KB tap
// image are the better fit), 4x4-tileable output, N % 4 == 0, one of the instantiated (n-tiles, k, stride) shapes
bool yl_dws_supported(int cin, int n, int dk, int ds, int oh, int ow) {
  const int kb = (cin + 15) / 16, ntt = (n + 15) / 16;
  if (kb < 12 || (n & 3) || (cin & 3) || (oh & 3) || (ow & 3) || oh < 4 || ow < 4) return false;
#define YL_DWS_CHECK(A, G, K, S_) if (dk == K && ds == S_ && ntt <= A * G && ntt > (A * G == 8 ? 6 : A * G == 13 ? 8 : 13)) return true;
  YL_DWS_SHAPES(YL_DWS_CHECK)
#undef YL_DWS_CHECK
  return false;
}
#endif

hipError_t yl_launch_conv_dws(const YlConvP& p, hipStream_t st) {
  if ((size_t)p.B * p.H * p.W * p.Cin * sizeof(yl_act_t) >= ((size_t)1 << 31)) return hipErrorNotSupported;   // 32-bit byte offsets
  if (p.dw_k == 0 || p.k != 1 || p.stride != 1 || p.dec_boxes || p.C1 > 0 || p.scale || p.in_shift || p.w3p ||
      !yl_dws_supported(p.Cin, p.N, p.dw_k, p.dw_stride, p.OH, p.OW) ||
      (size_t)p.B * p.H * p.W * p.Cin >= ((size_t)1 << 40))
    return hipErrorNotSupported;
  const int ntt = p.NTtot;
#define YL_DWS_RUN(A, G, K, S_) \
  if (p.dw_k == K && p.dw_stride == S_ && ntt <= A * G && ntt > (A * G == 8 ? 6 : A * G == 13 ? 8 : 13)) return dws_go<A, G, K, S_, 4>(p, st, false);
  YL_DWS_SHAPES(YL_DWS_RUN)
#undef YL_DWS_RUN
  return hipErrorNotSupported;
}

static hipError_t yl_dws_init() {
  YlConvP q = {};
  hipError_t e = hipSuccess;
#define YL_DWS_ATTR(A, G, K, S_) if (e == hipSuccess) e = dws_go<A, G, K, S_, 4>(q, nullptr, true);
  YL_DWS_SHAPES(YL_DWS_ATTR)
#undef YL_DWS_ATTR
  return e;
}

// ------------------------------------------------------------------------------------------------
// Dense 3x3 stride-1 convolution as Winograd F(2x2,3x3): Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A -- 16 multiplications
// per 2x2 output tile and channel pair instead of 36 (2.25x fewer MFMAs; option "winograd": every eligible layer by
// default since round 4 -- the transforms round differently from the direct convolution, so the result is not
// bit-identical to it, but the score error against the fp32 oracle was measured equal: profiles/r04_winograd_margin*.json).
// One wave owns 16 Winograd tiles (lane pl = tile, 4 channels 4kq..4kq+3 of a 16-channel block) and 2 n-tiles, and
// holds ALL 16 transform positions' accumulators (16 x 2 x 4 = 128 VGPRs), so the loop is channel-block outer:
//   per k-block:  4x4 input patch of the lane's tile (16 float4 from L1/L2, zero buffer outside the image)
//                 column transform in place (16 float4 ops), then per position xi one row-transform op -> B operand,
//                 two A fragments of U_xi from the LDS chunk, 8 MFMAs
// The U image ([n-group][k-block][xi][2][64][4], pack_wino in yl_api.hip) streams through LDS in 32 KiB chunks,
// double-buffered, one barrier per k-block, 8 waves (128 tiles = 512 output pixels) share a chunk; (n-group, m-tile)
// items group-major in XCD bands like yl_conv_kxk_kernel.  Output transform, bias, ReLU-family clamp and the four
// NHWC float4 stores per n-tile in the epilogue (ReLU-family clamp or SiLU).
// SH = 1 (round 4): the conv reads its input nearest-upsampled by 2 (yl_layer.in_shift, the prototype branch's second conv):
// p.H / p.W are the dims of the VIRTUAL tensor; the 4x4 patch of a tile maps onto a 3x3 block of stored pixels
// (rows / columns (2t-1+r) >> 1 = t-1, t, t, t+1).
template <int SH>
__global__ __launch_bounds__(512, 2) void yl_conv_wino_kernel(YlConvP p) {
  constexpr int NW = 8, NT = 2;
  extern __shared__ __attribute__((aligned(16))) float yl_clds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  const int KB = p.KB, Cin = p.Cin, H = p.H, W = p.W, OH = p.OH, OW = p.OW, N = p.N;
  const int TW = (OW + 1) >> 1, TH = (OH + 1) >> 1;
  const int timg = TW * TH;
  const long T = (long)p.B * timg;                            // Winograd tiles
  const yl_act_t* const xin = p.x;
  const long zdelta = p.zeros - p.x;
  f32x4* wl = reinterpret_cast<f32x4*>(yl_clds);            // [2][16][NT][64] float4
  const f32x4* wg = reinterpret_cast<const f32x4*>(p.wino);
  const int G = (p.NTtot + NT - 1) / NT;
  const int bx = blockIdx.x, gx = gridDim.x;                 // gx % 8 == 0
  const int per = gx >> 3, slot = bx >> 3;
  const int tpx = (p.ntiles + 7) >> 3;
  const int band0 = (bx & 7) * tpx;
  const int band1 = (band0 + tpx) < p.ntiles ? (band0 + tpx) : p.ntiles;
  const int bt = band1 > band0 ? band1 - band0 : 0;
  const int nitems = bt * G;
  const int nmine = slot < nitems ? (nitems - 1 - slot) / per + 1 : 0;
  const long total_chunks = (long)nmine * KB;
  auto load_chunk = [&](int g, int kb, int buf) {
    for (int i = wave; i < 16 * NT; i += NW)
      yl_glds16(wg + (((size_t)g * KB + kb) * 16 * NT + i) * 64 + lane, wl + ((size_t)buf * 16 * NT + i) * 64);
  };
  // item -> (n-group g, m-tile): n-groups in blocks of GBS, inside a block m-tile-major with the GBS groups innermost,
  // so the workgroups of an XCD (consecutive items) read the SAME 128-tile input window for GBS n-groups at a time
  // and stream only GBS / G of the U image: both stay in the XCD's 4 MB L2 (group-major order re-read the input
  // from HBM / Infinity Cache once per n-group: 11 x 269 MB per launch at 80x80)
  constexpr int GBS = 4;
  const int nbf = G / GBS;
  auto item_g = [&](int item, int& mt) {
    int gb = item / (bt * GBS), cnt = GBS;
    if (gb >= nbf) { gb = nbf; cnt = G - nbf * GBS; }
    const int rem = item - gb * bt * GBS;
    mt = rem / cnt;
    return gb * GBS + (rem - mt * cnt);
  };
  int mt0 = 0;
  if (total_chunks > 0) load_chunk(item_g(slot, mt0), 0, 0);
  long gchunk = 0;
  __syncthreads();
  // Two wave groups half a k-block out of phase.  Wave w of a workgroup runs on SIMD w % 4, so waves 0-3 and 4-7 are
  // one wave per SIMD each: while group 0 waits for the patch of block n + 1 and transforms it (P1), group 1 issues
  // the MFMAs of block n (P2) on the same SIMDs, then the roles swap.  Every wave runs  P1; barrier; P2; barrier  and
  // group 1 starts one barrier late; chunk n + 1 is copied to LDS in the slot between the two groups' uses of the
  // buffer it replaces (each wave its share).  In phase, all eight waves waited for memory together: smooth3 2.59 ms;
  // out of phase 2.06 ms.  (Grouping by w & 1 or (w >> 1) & 1 puts both waves of a SIMD in one group: 3.27 ms.)
  const int grp = wave >> 2;
  if (grp == 1) __syncthreads();
  // SH == 0 (round 6): the patch loads go through a raw buffer descriptor whose base sits (W + 1) pixels IN FRONT of the tensor, so
  // that the lane's part of every address -- the patch origin, which may lie one row and one column outside the image -- is ONE
  // non-negative 32-bit byte offset for the tile and the element / k-block part is a scalar offset; elements outside the image
  // take an out-of-range offset (one select each; the loop carried a select pair, a compare and a 64-bit add per element: 85
  // vector instructions per k-block for 16 loads).  The channel tail is not masked (U is zero there, pack_wino).
  const int borg = (W + 1) * Cin * (int)sizeof(yl_act_t);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(xin) - borg), 0, (int)((long)p.B * H * W * Cin * (long)sizeof(yl_act_t)) + borg, 0x00020000);
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;

  for (int wi = 0; wi < nmine; ++wi) {
    const int item = slot + wi * per;
    int mtl = 0;
    const int g = item_g(item, mtl);
    const int mtile = band0 + mtl;
    long t = ((long)mtile * NW + wave) * 16 + pl;
    const bool tvalid = t < T;
    if (!tvalid) t = T - 1;
    const int b = (int)(t / timg);
    const int trem = (int)(t - (long)b * timg);
    const int ty = trem / TW, tx = trem - ty * TW;
    // patch offsets (floats from p.x; the launcher guarantees the tensor is < 2^31 floats): 16 pixels, rows 2ty-1 ..
    // 2ty+2, columns 2tx-1 .. 2tx+2; bit e of `inb` clear = outside the image (zero buffer)
    const int pbase = ((b * H + 2 * ty - 1) * W + 2 * tx - 1) * Cin + 4 * kq;   // patch origin (may lie outside: masked)
    int rowoff[4], coloff[4];                                    // SH: stored-tensor offsets of the patch rows / columns
    if (SH) {
      const int Hs = H >> SH, Ws = W >> SH;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rowoff[r] = ((b * Hs + ((2 * ty - 1 + r) >> SH)) * Ws) * Cin + 4 * kq;
        coloff[r] = ((2 * tx - 1 + r) >> SH) * Cin;
      }
    }
    unsigned inb = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int iy = 2 * ty - 1 + r, ix = 2 * tx - 1 + q;
        inb |= (iy >= 0 && iy < H && ix >= 0 && ix < W) ? (1u << (r * 4 + q)) : 0u;
      }
    const int vbase = pbase * (int)sizeof(yl_act_t) + borg;       // >= 0: the patch origin seen from the descriptor's base
    auto load_row = [&](int kb, int r, f32x4 (&dst)[16]) {
      if (!SH && !defined_YL_F16S) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = r * 4 + q;
          dst[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ((inb >> e) & 1u) ? vbase : (int)0x80000000u,
                                                                                 ((r * W + q) * Cin + kb * 16) * (int)sizeof(yl_act_t), 0));
        }
        return;
      }
      const bool tail = kb * 16 + 4 * kq >= Cin;                  // channel tail of the last block: zeros
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = r * 4 + q;
        const int off = SH ? rowoff[r] + coloff[q] + kb * 16
                           : pbase + (r * W + q) * Cin + kb * 16;  // (r * W + q) * Cin + kb * 16: wave-uniform
        dst[e] = yl_ld4((((inb >> e) & 1u) && !tail) ? xin + off : xin + zdelta);
      }
    };
    f32x4 acc[16][1][NT];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[xi][0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < KB; ++kb, ++gchunk) {
      const int buf = (int)(gchunk & 1);
      const bool more = gchunk + 1 < total_chunks;
      int gn = g, kn = kb + 1;
      if (kn == KB) { int mtn = 0; kn = 0; gn = more ? item_g(item + per, mtn) : g; }
      if (grp == 1 && more) load_chunk(gn, kn, buf ^ 1);
      f32x4 d[16];
#pragma unroll
      for (int r = 0; r < 4; ++r) load_row(kb, r, d);
      // column transform B^T d (rows of B^T: d0-d2, d1+d2, d2-d1, d1-d3), in place per column
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 d0 = d[q], d1 = d[4 + q], d2 = d[8 + q], d3 = d[12 + q];
        d[q] = d0 - d2; d[4 + q] = d1 + d2; d[8 + q] = d2 - d1; d[12 + q] = d1 - d3;
      }
      __syncthreads();
      if (grp == 0 && more) load_chunk(gn, kn, buf ^ 1);
      const f32x4* wb = wl + (size_t)buf * 16 * NT * 64 + lane;
      // A fragments of position xi + 1 are requested before the MFMAs of position xi (explicit double buffer, order
      // pinned with sched_barrier: left alone the compiler issued the two ds_read_b128 right in front of the MFMAs
      // that need them and every position waited out an LDS round trip)
      f32x4 wq[2][NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wq[0][nt] = wb[nt * 64];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        const int i = xi >> 2, j = xi & 3;
        const f32x4 t0 = d[i * 4], t1 = d[i * 4 + 1], t2 = d[i * 4 + 2], t3 = d[i * 4 + 3];
        f32x4 xq[1];
        xq[0] = j == 0 ? t0 - t2 : j == 1 ? t1 + t2 : j == 2 ? t2 - t1 : t1 - t3;
        if (xi + 1 < 16) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) wq[(xi + 1) & 1][nt] = wb[((xi + 1) * NT + nt) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);                         // reads of position xi + 1 stay above ...
        yl_mma_step<NT, 1>(wq[xi & 1], xq, acc[xi]);
        __builtin_amdgcn_sched_barrier(0);                         // ... the MFMAs of position xi
      }
      if (kb + 1 < KB) __syncthreads();                            // (after the last block: behind the epilogue)
    }
    // output transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]), bias, clamp, store the 2x2 pixels
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = (g * NT + nt) * 16 + 4 * kq;
      f32x4 r0[4], r1[4];                                         // A^T M : two rows x four columns
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r0[j] = acc[j][0][nt] + acc[4 + j][0][nt] + acc[8 + j][0][nt];
        r1[j] = acc[4 + j][0][nt] - acc[8 + j][0][nt] - acc[12 + j][0][nt];
      }
      f32x4 y[2][2];
      y[0][0] = r0[0] + r0[1] + r0[2]; y[0][1] = r0[1] - r0[2] - r0[3];
      y[1][0] = r1[0] + r1[1] + r1[2]; y[1][1] = r1[1] - r1[2] - r1[3];
      if (tvalid && n < N) {
        const f32x4 bias = yl_ld4(p.bias + n);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            const int oy = 2 * ty + a, ox = 2 * tx + c2;
            if (oy < OH && ox < OW) {
              const size_t o = (((size_t)b * OH + oy) * OW + ox) * N + n;
              f32x4 v = yl_actc(y[a][c2] + bias, p.act, lo, hi);
              if (p.res) v += yl_ld4(p.res + o);                     // residual after the activation (yl_epi_generic's order)
              yl_st4(p.out + o, v);
            }
          }
      }
    }
    __syncthreads();               // closes the item's last P2 slot
  }
  if (grp == 0) __syncthreads();   // group 1's last slot
}

static hipError_t wino_go(const YlConvP& p0, hipStream_t st, bool attr_only) {
  if (attr_only) {
    const hipError_t e = hipFuncSetAttribute((const void*)yl_conv_wino_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)yl_conv_wino_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  }
  YlConvP p = p0;
  const long T = (long)p.B * ((p.OH + 1) >> 1) * ((p.OW + 1) >> 1);
  p.ntiles = (int)((T + 127) / 128);
  const size_t lds = (size_t)2 * 16 * 2 * 1024;
  const int res = yl_resident_blocks_n(yl_conv_wino_kernel<0>, 512, lds);
  const int G = (p.NTtot + 1) / 2;
  int gx = res & ~7;
  while (gx > 8 && gx - 8 >= p.ntiles * G) gx -= 8;
  if (p.in_shift) hipLaunchKernelGGL(yl_conv_wino_kernel<1>, dim3(gx), dim3(512), lds, st, p);
  else hipLaunchKernelGGL(yl_conv_wino_kernel<0>, dim3(gx), dim3(512), lds, st, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3), second form (round 5): the 16 transform positions are dealt to the WAVES -- wave w owns positions
// (i, j) = (w >> 1, 2 (w & 1) + {0, 1}) -- and every wave holds MT m-tiles x NT n-tiles of accumulators for its two
// positions.  yl_conv_wino_kernel gives each wave all 16 positions of 16 tiles x 2 n-tiles: per 128 MFMAs it fetches and
// transforms a 16 KB patch and reads 32 KB of U from LDS, the input is re-fetched and re-transformed once per 32 output
// channels (11 times at N = 328), and its two wave groups alternate between a memory phase and an MFMA phase with two
// barriers per k-block (MFMA pipe 56 % busy at 80 x 80).  Here a workgroup item is MT m-tiles (m-tile = 4 x 4 Winograd
// tiles = 8 x 8 output pixels of one image) x NT n-tiles; per 16-channel k-block:
//   window  the 10 x 10-pixel input window of each m-tile (64 B per pixel) lands in LDS by asynchronous LDS-DMA copies
//           (MT per wave, no VGPRs), double-buffered, requested a whole k-block ahead: every input pixel is fetched ONCE
//           per item -- not once per tile that touches it -- and once per NT n-tiles (7 times at N = 328, NT = 3)
//   B       a position (i, j) of B^T d B is a signed sum of FOUR window pixels (rows {0,2} {1,2} {2,1} {1,3} by i, columns
//           likewise by j), so a wave builds the B fragments of its two positions itself: per m-tile 6 ds_read_b128 (2 rows
//           x 3 columns), 3 row combinations X + s Y, 2 column combinations -- 10 packed VALU operations between the MFMAs.
//           No transformed image in LDS, no second barrier.  (A first version wrote V = B^T d B to LDS with all 512
//           threads and read it back as B fragments: two barriers per k-block, a 16-read / 8-write LDS round trip per
//           thread and a 160 KB footprint; tools/wino_stamps.py and the ablation builds priced that round trip at 0.24 ms
//           of a 1.98 ms launch and the bare MFMA + barrier skeleton at 1.45 ms.)
//   A       NT 1-KB fragment loads of U_xi straight from L1/L2 per position, requested one position ahead -- U of a
//           position is used by one wave only, LDS would not share anything
//   MFMA    2 x MT x NT x 4 per wave and k-block; the memory requests sit BETWEEN them (issued back to back -- 8 waves x 7
//           instructions at one moment -- they filled the CU's vector-memory queue, and in-order issue kept every wave in
//           front of its MFMAs until its own requests were taken: 2300 of 11000 cycles per k-block)
// ONE barrier per k-block.  After the K loop the accumulators of one n-tile at a time go through LDS so that thread
// (m-tile, lane, output row) gathers the 12 positions of its row: output transform, bias, activation, residual, float4
// NHWC stores.  Same transform expressions (X - Y == X + (-1) Y in one rounding), k order (k-blocks ascending, the four
// MFMAs of a block in yl_mma_step's order per accumulator) and epilogue as yl_conv_wino_kernel: BIT-IDENTICAL to it
// (tests/test_gpu_parity.py).  Reads the same U image (pack_wino).
// Window layout (16-byte slots, 512 per m-tile and buffer, 480 used): pixel index P = R(y) * 12 + C(x) with R(y) =
// 5 (y & 1) + (y >> 1), C likewise -- even rows / columns first, so that tiles two pixels apart are neighbours --,
// slot = 4 P + (kq ^ 2 ((P >> 2) & 1)).  (a) the four lanes of a quad copy the 64 contiguous bytes of ONE pixel: a copy
// instruction touches 16 cache lines, not 64; (b) a ds_read_b128 lane group holds 8 tiles with channel group kq = a
// (tile rows {0,3}) and 8 with a ^ 1 (tile rows {1,2}) (MI355X_MICROARCH.md, LDS): P = ty4 * 12 + tx4 + const takes every
// residue mod 4 twice per set, the two with bit 2 of P different (36 = 9 * 4, 12 = 3 * 4), so the low slot bits are
// {a, a ^ 2} and {a ^ 1, a ^ 3}: sixteen distinct 16-byte bank groups, conflict-free.
template <int MT, int NT, int SH>
__global__ __launch_bounds__(512, 2) void yl_conv_wino2_kernel(YlConvP p) {
  constexpr int RP = 12, RM = 512;                              // slots per m-tile window (480 used: 8 full copy instructions)
  extern __shared__ __attribute__((aligned(16))) float yl_clds[];
  f32x4* const Rl = reinterpret_cast<f32x4*>(yl_clds);          // [2][MT][RM] windows
  f32x4* const Xl = Rl + 2 * MT * RM;                           // [16][MT][64] accumulator exchange of the epilogue
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;                      // MFMA lane: 4 channels 4kq.. of tile pl
  const int ty4 = pl >> 2, tx4 = pl & 3;
  const int KB = p.KB, Cin = p.Cin, H = p.H, W = p.W, OH = p.OH, OW = p.OW, N = p.N;
  const int Hs = H >> SH, Ws = W >> SH;
  const int TW = (OW + 1) >> 1, TH = (OH + 1) >> 1;
  const int MX = (TW + 3) >> 2, MY = (TH + 3) >> 2;
  const int mimg = MX * MY;
  const long MTOT = (long)p.B * mimg;                           // m-tiles
  const yl_act_t* const xin = p.x;
  const int NG2 = (p.NTtot + 1) >> 1;                           // n-tile pairs of the U image
  const int G = (p.NTtot + NT - 1) / NT;
  const int bx = blockIdx.x, gx = gridDim.x;                    // gx % 8 == 0
  const int per = gx >> 3, slot = bx >> 3;
  const int tpx = (p.ntiles + 7) >> 3;                          // m-blocks per XCD band
  const int band0 = (bx & 7) * tpx;
  const int band1 = (band0 + tpx) < p.ntiles ? (band0 + tpx) : p.ntiles;
  const int bt = band1 > band0 ? band1 - band0 : 0;
  const int nitems = bt * G;
  const int nmine = slot < nitems ? (nitems - 1 - slot) / per + 1 : 0;
  // item -> (n-group, m-block): as in yl_conv_wino_kernel (the workgroups of an XCD walk GBS n-groups of one input window)
  constexpr int GBS = 4;
  const int nbf = G / GBS;
  auto item_g = [&](int item, int& mt) {
    int gb = item / (bt * GBS), cnt = GBS;
    if (gb >= nbf) { gb = nbf; cnt = G - nbf * GBS; }
    const int rem = item - gb * bt * GBS;
    mt = rem / cnt;
    return gb * GBS + (rem - mt * cnt);
  };
  // copy role of the lane: slot rs = wave * 64 + lane of every m-tile's window
  const int rs = wave * 64 + lane;
  const int rP = rs >> 2, rkq = (rs & 3) ^ (((rP >> 2) & 1) << 1);
  const int rpr = rP / RP, rpc = rP - rpr * RP;
  const int ry = rpr < 5 ? 2 * rpr : 2 * (rpr - 5) + 1, rx = rpc < 5 ? 2 * rpc : 2 * (rpc - 5) + 1;
  const bool rpix = rpr < 10 && rpc < 10;                        // (slots 480..511 and the pad columns copy zeros)
  // position role of the wave: rows X + sr * Y, columns u0 - u1 and u1 + sc * u2
  const int pi = wave >> 1, pj = wave & 1;
  const int rowX = pi == 0 ? 0 : pi == 2 ? 2 : 1, rowY = pi == 2 ? 1 : pi == 3 ? 3 : 2;
  const float sr = pi == 1 ? 1.0f : -1.0f, sc = pj ? -1.0f : 1.0f;
  const f32x4 sr4 = (f32x4){sr, sr, sr, sr}, sc4 = (f32x4){sc, sc, sc, sc};
  const int col0 = pj ? 2 : 0, col1 = pj ? 1 : 2, col2 = pj ? 3 : 1;
  auto wslot = [&](int r, int c) {                               // the lane's slot of window pixel (2 ty4 + r, 2 tx4 + c)
    const int P = (ty4 + (r & 1) * 5 + (r >> 1)) * RP + tx4 + (c & 1) * 5 + (c >> 1);
    return 4 * P + (kq ^ (((P >> 2) & 1) << 1));
  };
  const int sX0 = wslot(rowX, col0), sX1 = wslot(rowX, col1), sX2 = wslot(rowX, col2);
  const int sY0 = wslot(rowY, col0), sY1 = wslot(rowY, col1), sY2 = wslot(rowY, col2);
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;

  // Both operand streams go through raw buffer descriptors (round 6): the per-lane part of an address is ONE 32-bit byte
  // offset fixed for the whole item, the k-block part is a scalar offset -- no vector instruction in the loop computes an
  // address (the loop carried 14 64-bit adds and 8 selects for its ten requests).  Window lanes outside the image carry an
  // offset beyond num_records: the hardware range check makes the copy write zeros.  The channel tail of the last k-block is
  // not masked: those lanes copy the first channels of the next pixel (finite values; the context's arenas end in 256 spare
  // bytes) and U is zero there (pack_wino pads with zeros), so the products are zeros as before.
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<yl_act_t*>(xin), 0, (int)((long)p.B * Hs * Ws * Cin * (long)sizeof(yl_act_t)), 0x00020000);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.wino), 0, (int)((long)NG2 * KB * 16 * 2 * 1024), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int lane16 = lane * 16;
  unsigned voff[MT];                                             // byte offset of the lane's window slot (k-block 0), OOB = zeros
  auto setup = [&](int mblock) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const long mi = (long)mblock * MT + mt;
      const int b = (int)(mi / mimg);
      const int r = (int)(mi - (long)b * mimg);
      const int my = r / MX, mx = r - my * MX;
      const int gy = 8 * my - 1 + ry, gxx = 8 * mx - 1 + rx;
      const bool in = rpix && mi < MTOT && gy >= 0 && gy < H && gxx >= 0 && gxx < W;
      voff[mt] = in ? (unsigned)((((b * Hs + (gy >> SH)) * Ws + (gxx >> SH)) * Cin + 4 * rkq) * (int)sizeof(yl_act_t)) : OOB;
    }
  };
  auto issue_raw = [&](int kb, int buf) {                        // branch-free: it is scheduled between MFMAs
#if defined(YL_WINO_ABL) && (YL_WINO_ABL & 1)                    // ablation builds (tools/run_wino_abl.sh): results wrong
    if (kb > 0) return;
#endif
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(Rl + (buf * MT + mt) * RM + wave * 64), 16,
                                               (int)voff[mt], kb * 16 * (int)sizeof(yl_act_t), 0, 0);
  };
  // Parity: the loop below is unrolled by two so that the window buffer and the U register set of a block are compile-time
  // (LDS read offsets become immediates, no register renaming at the block's end).  Block kb uses buffer / set
  // (kb + KB) & 1: the LAST block always sits in buffer 1, an odd KB starts with one single block out of buffer 1.
  const int sh = KB & 1;

  int mb0 = 0, g = 0;
  if (nmine > 0) { g = item_g(slot, mb0); setup(band0 + mb0); issue_raw(0, sh); }
  for (int wi = 0; wi < nmine; ++wi) {
    const int mblock = band0 + mb0;
    // U fragment of (n-tile g*NT + nt, k-block kb, position xi): pair-major image of pack_wino; scalar byte offset
    int ubs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int ntg = g * NT + nt;
      if (ntg >= 2 * NG2) ntg = 0;                                // beyond the image: any fragment (columns never stored)
      ubs[nt] = (((ntg >> 1) * KB * 32 + (ntg & 1)) * 64 + 2 * wave * 128) * 16;
    }
    auto load_u = [&](int kb, int ps, f32x4 (&dst)[NT]) {
#if defined(YL_WINO_ABL) && (YL_WINO_ABL & 2)
      if (kb > 0) return;
#endif
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        dst[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, lane16, ubs[nt] + (kb * 16 + ps) * 2048, 0));
    };
    f32x4 acc[2][MT][NT];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[ps][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // B fragments of the wave's two positions for m-tile mt out of window buffer `buf`
    auto load_win = [&](int buf, int mt, f32x4 (&x)[3], f32x4 (&y)[3]) {
      const f32x4* const wb = Rl + (buf * MT + mt) * RM;
      x[0] = wb[sX0]; x[1] = wb[sX1]; x[2] = wb[sX2];
      y[0] = wb[sY0]; y[1] = wb[sY1]; y[2] = wb[sY2];
    };
    auto make_b = [&](const f32x4 (&x)[3], const f32x4 (&y)[3], f32x4& b0, f32x4& b1) {
#if defined(YL_WINO_PK) && !YL_WINO_PK                               // (A/B builds)
      const f32x4 u0 = y[0] * sr4 + x[0], u1 = y[1] * sr4 + x[1], u2 = y[2] * sr4 + x[2];
      b0 = u0 - u1;
      b1 = u2 * sc4 + u1;
#else
      yl_wino_b(x, y, sr4, sc4, b0, b1);
#endif
    };
    auto mma_mt = [&](const f32x4 (&a)[NT], const f32x4& b, f32x4 (&c)[NT]) {    // one m-tile: the 4 steps x NT n-tiles
#if YL_BF16
      const f32x4 bb[1] = {b};
      f32x4 cc[1][NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) cc[0][nt] = c[nt];
      yl_mma_step<NT, 1>(a, bb, cc);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) c[nt] = cc[0][nt];
#else
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) c[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt][st], b[st], c[nt], 0, 0, 0);
#endif
    };
    // One k-block: MT x (window reads of the next m-tile, B fragments, 2 x NT x 4 MFMAs).  The block's ONE barrier sits in
    // front of the LAST m-tile's MFMAs: by then the wave has read all it needs of window(kb) and its copies of
    // window(kb + 1) (requested a k-block ago) have landed, so behind the barrier window(kb + 1) is complete and window(kb)'s
    // buffer is free -- the wave requests window(kb + 2) into it and the first m-tile's reads of block kb + 1, and only
    // then issues the last m-tile's 2 x NT x 4 MFMAs: barrier and LDS latency lie under them and the MFMA stream runs on
    // across k-blocks (with the barrier at the top of the block both waves of a SIMD met it with nothing to issue).
    // MODE 0: any block; 1: the second-to-last (nothing left to request); 2: the last (peeled like this so that the body
    // is branch-free: with conditional requests every merge waited for all outstanding loads).  The U fragments of block
    // kb + 1 are requested into the OTHER register set under the first two m-tiles -- requested behind the last use of this
    // block's set they had no time to arrive.  PAR = the block's window buffer and U set.
    f32x4 ua[2][2][NT];                                           // [set][position][n-tile]
    f32x4 x[2][3], y[2][3];
    auto kblock = [&](int kb, auto par, auto mode) {
      constexpr int PAR = decltype(par)::value;
      constexpr int MODE = decltype(mode)::value;
      WINO_STAMP(kb * 7 + 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (mt + 1 < MT) load_win(PAR, mt + 1, x[(mt + 1) & 1], y[(mt + 1) & 1]);
        f32x4 b0, b1;
        make_b(x[mt & 1], y[mt & 1], b0, b1);
        if (mt == MT - 1 && MODE < 2) {
          __builtin_amdgcn_sched_barrier(0);
          WINO_STAMP(kb * 7 + 4);
          __builtin_amdgcn_s_waitcnt(0x0070);                     // vmcnt(0) lgkmcnt(0): the wave's own window copies and reads (not left to the
          __syncthreads();                                        // compiler: LDS-DMA requests of the previous iteration are not carried over the back edge)
          WINO_STAMP(kb * 7 + 5);
          if (MODE == 0) issue_raw(kb + 2, PAR);
          load_win(PAR ^ 1, 0, x[0], y[0]);
          __builtin_amdgcn_sched_barrier(0);
          WINO_STAMP(kb * 7 + 6);
        }
        mma_mt(ua[PAR][0], b0, acc[0][mt]);
        if (MODE < 2 && mt == 0) load_u(kb + 1, 0, ua[PAR ^ 1][0]);
        mma_mt(ua[PAR][1], b1, acc[1][mt]);
        if (MODE < 2 && mt == (MT > 2 ? 1 : 0)) load_u(kb + 1, 1, ua[PAR ^ 1][1]);
        // m-tile by m-tile: left alone the scheduler gathered all window reads at the top (spills) and pushed the U
        // requests behind the last MFMA
        __builtin_amdgcn_sched_barrier(0);
        if (mt < 3) WINO_STAMP(kb * 7 + 1 + mt);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    int kb = 0;
    if (sh) {                                                     // odd KB: block 0 on its own, out of buffer / set 1
      load_u(0, 0, ua[1][0]);
      load_u(0, 1, ua[1][1]);
      __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
      __syncthreads();                                            // window(0) landed (requested under the previous epilogue)
      issue_raw(1, 0);
      load_win(1, 0, x[0], y[0]);
      kblock(0, I1{}, I0{});
      kb = 1;
    } else {
      load_u(0, 0, ua[0][0]);
      load_u(0, 1, ua[0][1]);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      issue_raw(1, 1);
      load_win(0, 0, x[0], y[0]);
    }
    for (; kb + 3 < KB; kb += 2) {
      kblock(kb, I0{}, I0{});
      kblock(kb + 1, I1{}, I0{});
    }
    kblock(KB - 2, I0{}, I1{});
    kblock(KB - 1, I1{}, I2{});
    // the item's coordinates for the epilogue, then the next item's window(0) request flies under the epilogue
    const int eg = g;
    const int emt = wave & (MT - 1), epart = wave / MT;            // output role: m-tile, row a (and column c2 when MT = 2)
    const long emi = (long)mblock * MT + emt;
    if (wi + 1 < nmine) { g = item_g(slot + (wi + 1) * per, mb0); setup(band0 + mb0); }
    const bool evalid = emi < MTOT;
    const int eb = (int)((evalid ? emi : 0) / mimg);
    const int er = (int)((evalid ? emi : 0) - (long)eb * mimg);
    const int emy = er / MX, emx = er - emy * MX;
    const int oa = epart & 1;
    const int oy = 2 * (4 * emy + ty4) + oa, ox0 = 2 * (4 * emx + tx4);
    __syncthreads();                                              // every wave is done with the windows
    if (wi + 1 < nmine) issue_raw(0, sh);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt > 0) __syncthreads();                                // the previous n-tile's exchange is read
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Xl[((size_t)(2 * wave + ps) * MT + mt) * 64 + lane] = acc[ps][mt][nt];
      __syncthreads();
      const f32x4* const mb = Xl + (size_t)emt * 64 + lane;
      f32x4 r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 m1 = mb[((4 + j) * MT) * 64], m2 = mb[((8 + j) * MT) * 64];
        if (oa == 0) r[j] = mb[(j * MT) * 64] + m1 + m2;
        else r[j] = m1 - m2 - mb[((12 + j) * MT) * 64];
      }
      f32x4 yy[2];
      yy[0] = r[0] + r[1] + r[2]; yy[1] = r[1] - r[2] - r[3];
      const int n = ((eg * NT + nt) * 16) + 4 * kq;
      if (evalid && n < N && oy < OH) {
        const f32x4 bias = yl_ld4(p.bias + n);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          if (MT == 2 && c2 != (epart >> 1)) continue;
          const int ox = ox0 + c2;
          if (ox < OW) {
            const size_t o = (((size_t)eb * OH + oy) * OW + ox) * N + n;
            f32x4 v = yl_actc(yy[c2] + bias, p.act, lo, hi);
            if (p.res) v += yl_ld4(p.res + o);
            yl_st4(p.out + o, v);
          }
        }
      }
    }
  }
}

template <int MT, int NT>
static hipError_t wino2_go(const YlConvP& p0, hipStream_t st, bool attr_only) {
  const size_t lds = ((size_t)2 * MT * 512 + (size_t)16 * MT * 64) * 16;      // two window buffers + the epilogue exchange
  if (attr_only) {
    const hipError_t e = hipFuncSetAttribute((const void*)yl_conv_wino2_kernel<MT, NT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)yl_conv_wino2_kernel<MT, NT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  YlConvP p = p0;
  const int TW = (p.OW + 1) >> 1, TH = (p.OH + 1) >> 1;
  const long MTOT = (long)p.B * ((TW + 3) >> 2) * ((TH + 3) >> 2);
  p.ntiles = (int)((MTOT + MT - 1) / MT);
  const int res = yl_resident_blocks_n(yl_conv_wino2_kernel<MT, NT, 0>, 512, lds);
  const int G = (p.NTtot + NT - 1) / NT;
  int gx = res & ~7;
  if (gx < 8) gx = 8;
  while (gx > 8 && gx - 8 >= p.ntiles * G) gx -= 8;
  if (p.in_shift) hipLaunchKernelGGL((yl_conv_wino2_kernel<MT, NT, 1>), dim3(gx), dim3(512), lds, st, p);
  else hipLaunchKernelGGL((yl_conv_wino2_kernel<MT, NT, 0>), dim3(gx), dim3(512), lds, st, p);
  return hipGetLastError();
}

// dense 3x3 stride-1 pad-1 layers that carry a Winograd image (yl_api.hip builds it for >= 64 channels, plain
// ReLU-family epilogue; layer_params hands it over only under option "winograd").
hipError_t yl_launch_conv_wino(const YlConvP& p, hipStream_t st) {
  if (!p.wino || p.k != 3 || p.stride != 1 || p.dw_k > 0 || p.up || p.dec_boxes || p.C1 > 0 || (p.N & 3) || p.w3p || p.scale ||
      p.in_shift > 1 || ((size_t)p.B * (p.H >> p.in_shift) * (p.W >> p.in_shift) * p.Cin + (size_t)(p.W + 1) * p.Cin) * sizeof(yl_act_t) >= ((size_t)1 << 31))
    return hipErrorNotSupported;                            // (32-bit byte offsets: both forms read through buffer descriptors)
  // second form (positions across the waves): K loops long enough to amortise the accumulator exchange, grids that fill
  // the 4 x 4-tile m-tiles; "dev_select" bit 11 keeps the first form (bitwise A/B), bits 12-13 pick a shape (A/B runs)
  // (second form: 32-bit byte offsets into the input tensor and the U image)
  const bool small = (size_t)p.B * (p.H >> p.in_shift) * (p.W >> p.in_shift) * p.Cin * sizeof(yl_act_t) < ((size_t)1 << 31) &&
                     (size_t)((p.NTtot + 1) / 2) * p.KB * 32768 < ((size_t)1 << 31);
  if (!(p.dev & YL_DEV_WINO_V1) && p.KB >= 2 && p.NTtot >= 3 && small) {   // (KB >= 2: the peeled last two blocks)
    const int TW = (p.OW + 1) >> 1, TH = (p.OH + 1) >> 1;
    const int MX = (TW + 3) >> 2, MY = (TH + 3) >> 2;
    const long MTOT = (long)p.B * MX * MY;
    if ((long)TW * TH * 100 >= (long)MX * MY * 16 * 65) {         // >= 65 % of the m-tiles' Winograd tiles exist (20 x 20: 69 %, 0.232 -> 0.205 ms)
      const int pad3 = (p.NTtot + 2) / 3 * 3, pad4 = (p.NTtot + 3) / 4 * 4;
      int nt = pad3 <= pad4 ? 3 : 4;
      // 4 m-tiles halve the U bytes per MFMA (80 x 80: 1.79 against 1.95 ms) when there are >= 3 items per CU; with 4
      // n-tiles that shape spills (2 x 4 x 4 accumulator quads + two U register sets)
      int mt = nt == 3 && (MTOT / 4) * ((p.NTtot + nt - 1) / nt) >= 3 * YL_NUM_CU ? 4 : 2;
      const unsigned v = YL_DEV_WINO_SHAPE(p.dev);
      if (v == 1) { mt = 4; nt = 4; } else if (v == 2) { mt = 2; nt = 7; } else if (v == 3) { mt = 2; }
      if (mt == 4 && nt == 3) return wino2_go<4, 3>(p, st, false);
      if (mt == 4 && nt == 4) return wino2_go<4, 4>(p, st, false);
      if (mt == 2 && nt == 3) return wino2_go<2, 3>(p, st, false);
      if (mt == 2 && nt == 4) return wino2_go<2, 4>(p, st, false);
      if (mt == 2 && nt == 7) return wino2_go<2, 7>(p, st, false);
    }
  }
  return wino_go(p, st, false);
}

static hipError_t yl_wino2_init() {
  YlConvP q = {};
  hipError_t e = wino2_go<4, 3>(q, nullptr, true);
  if (e == hipSuccess) e = wino2_go<4, 4>(q, nullptr, true);
  if (e == hipSuccess) e = wino2_go<2, 3>(q, nullptr, true);
  if (e == hipSuccess) e = wino2_go<2, 4>(q, nullptr, true);
  if (e == hipSuccess) e = wino2_go<2, 7>(q, nullptr, true);
  return e;
}

// ------------------------------------------------------------------------------------------------
namespace {

template <typename K>
int yl_resident_blocks_c(K kernel, size_t lds) {
  static std::mutex mu;
  static std::map<std::pair<const void*, size_t>, int> cache;
  const std::pair<const void*, size_t> key((const void*)kernel, lds);
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kernel, 512, lds) != hipSuccess || nb < 1) nb = 1;
  if (nb > 4) nb = 4;
  cache[key] = nb * YL_NUM_CU;
  return nb * YL_NUM_CU;
}

template <int DK, int DS, int NTW>
hipError_t dwc_one(const YlConvMulti& m, int gx, size_t lds, hipStream_t st, bool attr_only, int* resident) {
  if (attr_only)
    return hipFuncSetAttribute((const void*)yl_conv_dwc_kernel<DK, DS, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               YL_DWC_LDS_MAX);
  if (resident) { *resident = yl_resident_blocks_c(yl_conv_dwc_kernel<DK, DS, NTW>, lds); return hipSuccess; }
  hipLaunchKernelGGL((yl_conv_dwc_kernel<DK, DS, NTW>), dim3(gx), dim3(512), lds, st, m);
  return hipGetLastError();
}

template <int NTW>
hipError_t dwc_dk(const YlConvMulti& m, int dk, int ds, int gx, size_t lds, hipStream_t st, bool attr_only,
                  int* resident) {
  hipError_t e = hipSuccess;
  if (attr_only || (dk == 3 && ds == 1)) { e = dwc_one<3, 1, NTW>(m, gx, lds, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || (dk == 3 && ds == 2)) { e = dwc_one<3, 2, NTW>(m, gx, lds, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || (dk == 5 && ds == 1)) { e = dwc_one<5, 1, NTW>(m, gx, lds, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || (dk == 5 && ds == 2)) { e = dwc_one<5, 2, NTW>(m, gx, lds, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  return attr_only ? hipSuccess : hipErrorInvalidValue;
}

hipError_t dwc_any(const YlConvMulti& m, int ntw, int dk, int ds, int gx, size_t lds, hipStream_t st,
                   bool attr_only, int* resident) {
  hipError_t e = hipSuccess;
  if (attr_only || ntw == 1) { e = dwc_dk<1>(m, dk, ds, gx, lds, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || ntw == 2) { e = dwc_dk<2>(m, dk, ds, gx, lds, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || ntw == 3) { e = dwc_dk<3>(m, dk, ds, gx, lds, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || ntw == 4) { e = dwc_dk<4>(m, dk, ds, gx, lds, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || ntw == 5) { e = dwc_dk<5>(m, dk, ds, gx, lds, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  return attr_only ? hipSuccess : hipErrorInvalidValue;
}

int kbmax_of(int ntw) { return ntw == 1 ? 18 : ntw == 2 ? 9 : ntw == 3 ? 6 : 4; }

}  // namespace

hipError_t yl_convc_init() {
  YlConvMulti m = {};
  YlConvP q = {};
  hipError_t e = kxk_go<7, 1, 4>(q, 1, nullptr, true);
  if (e == hipSuccess) e = kxk_go<7, 2, 4>(q, 1, nullptr, true);
  if (e == hipSuccess) e = kxk_go<7, 1, 8>(q, 1, nullptr, true);
  if (e == hipSuccess) e = kxk_go<4, 1, 8>(q, 1, nullptr, true);
  if (e == hipSuccess) e = kxk_go<4, 2, 4>(q, 1, nullptr, true);
  if (e == hipSuccess) e = yl_ir_init();
  if (e == hipSuccess) e = pws_nw<6>(q, nullptr, false, true);
  if (e == hipSuccess) e = pws_nw<7>(q, nullptr, false, true);
  if (e == hipSuccess) e = pws_nw<8>(q, nullptr, false, true);
  if (e == hipSuccess) e = pws_nw<9>(q, nullptr, false, true);
  if (e == hipSuccess) e = pws_nw<11>(q, nullptr, false, true);
  if (e == hipSuccess) e = pws_nw<13>(q, nullptr, false, true);
  if (e == hipSuccess) e = dwt_any(m, 0, nullptr, false, false, true);
  if (e == hipSuccess) e = dwt_splitk(m, nullptr, true);
  if (e == hipSuccess) e = wino_go(q, nullptr, true);
  if (e == hipSuccess) e = yl_wino2_init();
  if (e == hipSuccess) e = dwk_go<7, 1, 4>(q, nullptr, true);
  if (e == hipSuccess) e = dwk_go<7, 3, 4>(q, nullptr, true);
  if (e == hipSuccess) e = dwk_go<8, 1, 4>(q, nullptr, true);
  if (e == hipSuccess) e = dwk_go<8, 2, 4>(q, nullptr, true);
  if (e == hipSuccess) e = dwl_go<16>(q, nullptr, true);
  if (e == hipSuccess) e = dwl_go<21>(q, nullptr, true);
  if (e == hipSuccess) e = yl_dws_init();
  if (e != hipSuccess) return e;
  return dwc_any(m, 0, 0, 0, 0, 0, nullptr, true, nullptr);
}

// n problems of identical configuration (m.p[0..n-1] filled like for yl_conv_dwh_kernel).  Returns
// hipErrorNotSupported when the shape is outside this kernel's limits (the caller then takes the halo kernel).
hipError_t yl_launch_conv_dwc(YlConvMulti& m, hipStream_t st) {
  const YlConvP& p = m.p[0];
  const int ntw = (p.NTtot + 3) / 4;
  if (ntw < 1 || ntw > 5 || p.KB > kbmax_of(ntw) || (p.N & 3)) return hipErrorNotSupported;
  // Measured per layer (edge_n, B = 64, eager): the producer / consumer split pays where the depthwise phase is
  // long -- K >= 192 channels on a stride-1 depthwise (28 vs 38 us for 3x3, 44 vs 50 us for 5x5 at 20x20) -- and
  // loses on the short-K layers, where two or three of the four depthwise waves idle and the per-tile barrier costs
  // more than the weight prologue it replaces.  "dev_select" bit 3 lifts the restriction (A/B runs).
  const bool all = (p.dev & YL_DEV_DWC_ALL) != 0;       // "dev_select" bit 3: the bitwise-equivalence test
  if (!all && (p.KB < 12 || p.dw_stride != 1)) return hipErrorNotSupported;
  if (!((p.dw_k == 3 || p.dw_k == 5) && (p.dw_stride == 1 || p.dw_stride == 2))) return hipErrorNotSupported;
  const int HP = 3 * p.dw_stride + p.dw_k;
  const int NG = (p.KB + 3) / 4;
  const int PITCH = p.dw_stride == 1 ? HP * 32 + ((HP * 128) % 256 == 128 ? 0 : 32) : HP * 32 + 16;   // YlDwcGeo
  const size_t lds = ((size_t)2 * p.KB * 256 + (((size_t)(p.dw_k * p.dw_k + 1) * p.Cin + 3) & ~(size_t)3) +
                      (size_t)(NG < 4 ? NG : 4) * HP * PITCH) * 4;
  if (lds > YL_DWC_LDS_MAX) return hipErrorNotSupported;
  long tiles[4], total = 0;
  for (int k = 0; k < m.n; ++k) {
    if ((m.p[k].OH & 3) || (m.p[k].OW & 3) || (size_t)m.p[k].B * m.p[k].H * m.p[k].W * m.p[k].Cin * 4 >= ((size_t)1 << 31))
      return hipErrorNotSupported;
    tiles[k] = (long)m.p[k].B * (m.p[k].OH >> 2) * (m.p[k].OW >> 2);
    total += tiles[k];
  }
  int res = 0;
  hipError_t e = dwc_any(m, ntw, p.dw_k, p.dw_stride, 0, lds, st, false, &res);
  if (e != hipSuccess) return e;
  // grid: a multiple of 8 workgroups (XCD-aware tile bands, see the kernel), at most what is co-resident
  long gx = (total + 7) & ~7L;
  if (gx > res) gx = res & ~7L;
  if (gx < 8) gx = 8;
  if (m.n == 1) { m.p[0].blk0 = 0; m.p[0].nblk = 0; }
  else {
    int at = 0;
    for (int k = 0; k < m.n; ++k) {
      long nb = ((tiles[k] * gx + total / 2) / total + 7) & ~7L;
      if (nb < 8) nb = 8;
      m.p[k].blk0 = at;
      m.p[k].nblk = (int)nb;
      at += (int)nb;
    }
    gx = at;
  }
  return dwc_any(m, ntw, p.dw_k, p.dw_stride, (int)gx, lds, st, false, nullptr);
}
