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
// conv kernel (tests/test_gpu_parity.py: test_dwc_kernel_is_bitwise_the_halo_kernel).
//
// Limits (launcher falls back to yl_conv_dwh_kernel otherwise): OH, OW multiples of 4, N % 4 == 0,
// NTW = ceil(ceil(N/16)/4) <= 5 and ceil(Cin/16) <= KBMAX(NTW) (the register budget of the resident weights).
#if defined(YL_BF16) && YL_BF16
#define yl_conv_dwc_kernel yl_conv_dwc_kernel_bf16
#define yl_launch_conv_dwc yl_launch_conv_dwc_bf16
#define yl_convc_init yl_convc_init_bf16
#endif
#include <map>
#include <mutex>
#include <utility>
#include "yl_internal.h"
#include "yl_dev.h"
#include "yl_epi.h"

#define YL_DWC_LDS_MAX (150 * 1024)

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

template <int DK, int DS, int NTW>
__global__ __launch_bounds__(256, 2) void yl_conv_dwc_kernel(YlConvMulti mp, int dbuf) {
  YL_SELECT_PROBLEM_C(mp)
  constexpr int KBMAX = YlDwcCfg<NTW>::KBMAX;
  constexpr int HP = 3 * DS + DK;                         // halo edge in pixels
  constexpr int PITCHF = ((HP * 16 + 7) / 64) * 64 + 56;  // see yl_conv_dwh_kernel: conflict-free tap reads
  constexpr int HF4 = HP * HP * 4;                        // float4 elements of one halo patch (16 channels)
  constexpr int NSLOT = (HF4 + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) float yl_clds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, pl = lane & 15;
  const int KB = p.KB;
  // LDS carve: [2 or 1][KB][64] float4 B fragments | [DK*DK][Cin] taps, [Cin] bias | 4 halo regions
  f32x4* bbuf = reinterpret_cast<f32x4*>(yl_clds);
  float* dwl = yl_clds + (size_t)(dbuf ? 2 : 1) * KB * 256;
  float* halo = dwl + (((size_t)(DK * DK + 1) * p.Cin + 3) & ~(size_t)3) + wave * (HP * PITCHF);
  const float lo = (p.act == YL_ACT_RELU || p.act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float hi = (p.act == YL_ACT_RELU6) ? 6.0f : INFINITY;
  const float dlo = (p.dw_act == YL_ACT_RELU || p.dw_act == YL_ACT_RELU6) ? 0.0f : -INFINITY;
  const float dhi = (p.dw_act == YL_ACT_RELU6) ? 6.0f : INFINITY;

  const int tw = p.OW >> 2, th = p.OH >> 2;
  const int tiles_img = tw * th;
  const int ntiles = p.B * tiles_img;

  // lane constants of the halo staging (as yl_conv_dwh_kernel)
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
  const int rbase = ((pl >> 2) * DS) * PITCHF + ((pl & 3) * DS) * 16 + 4 * kq;
  constexpr unsigned OOB = 0x80000000u;
  unsigned goff[NSLOT];
  auto tile_geom = [&](int tile) {
    const int b = tile / tiles_img;
    const int trem = tile - b * tiles_img;
    const int tyi = trem / tw, txi = trem - tyi * tw;
    const int iy0 = 4 * tyi * DS - p.dw_pad_t, ix0 = 4 * txi * DS - p.dw_pad_l;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int e = j * 64 + lane;
      const int hp = (e < HF4 ? e : 0) >> 2;
      const int hr = hp / HP, hc = hp - hr * HP;
      const int iy = iy0 + hr, ix = ix0 + hc;
      const bool in = e < HF4 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      goff[j] = in ? (unsigned)((((b * p.H + iy) * p.W + ix) * p.Cin + (lane & 3) * 4) * 4) : OOB;
    }
  };
  auto stage_load = [&](int kb, f32x4 (&r)[NSLOT]) {
    const bool cok = kb * 16 + (lane & 3) * 4 < p.Cin;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const unsigned off = cok ? goff[j] + (unsigned)kb * 64u : OOB;
      r[j] = yl_ld4(off < OOB ? p.x + (off >> 2) : p.zeros);
    }
  };
  auto stage_store = [&](const f32x4 (&r)[NSLOT]) {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
      if (s_ok[j]) *reinterpret_cast<f32x4*>(halo + s_lo[j]) = r[j];
  };

  // ---- once per workgroup: first halo request, depthwise taps -> LDS (asynchronous), this wave's 1x1 weights -> registers
  f32x4 stg[NSLOT];
  int tile = bx;
  const bool p1 = wave < KB;                               // this wave has phase-1 work at all
  if (tile < ntiles && p1) {
    tile_geom(tile);
    stage_load(wave, stg);
  }
  {
    const int nw = DK * DK * p.Cin;
    yl_glds_floats(p.dw_w, dwl, nw, tid, 256);
    if (p.dw_b) yl_glds_floats(p.dw_b, dwl + nw, p.Cin, tid, 256);
    else for (int i = tid; i < p.Cin; i += 256) dwl[nw + i] = 0.0f;
  }
  const int nt0 = wave * NTW;
  const bool p2 = nt0 < p.NTtot;                           // this wave owns output channels
  f32x4 wreg[KBMAX][NTW];
  {
    const f32x4* wg = reinterpret_cast<const f32x4*>(p.wp);
#pragma unroll
    for (int kb = 0; kb < KBMAX; ++kb)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const bool ok = kb < KB && nt0 + nt < p.NTtot;
        wreg[kb][nt] = ok ? wg[((size_t)kb * p.NTtot + nt0 + nt) * 64 + lane] : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
  }
  const bool pre_add = p.res != nullptr && p.up == nullptr && p.act == YL_ACT_NONE;
  __syncthreads();                                         // taps are in LDS
  if (tile < ntiles && p1) stage_store(stg);

  int cur = 0;
  for (; tile < ntiles; tile += gx) {
    const int ntile = tile + gx;
    f32x4* bb = bbuf + (size_t)cur * KB * 64;
    // ---- phase 1: depthwise on this wave's channel blocks -> shared B fragments
    for (int kb = wave; kb < KB; kb += 4) {
      const bool more = kb + 4 < KB;
      const bool nxt = !more && ntile < ntiles;
      if (more) stage_load(kb + 4, stg);
      else if (nxt) { tile_geom(ntile); stage_load(wave, stg); }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // halo writes (all lanes) -> tap reads
      const int c = kb * 16 + 4 * kq;
      const int cs = c < p.Cin ? c : p.Cin - 4;
      f32x4 s = yl_ld4(dwl + DK * DK * p.Cin + cs);
      if (DK == 3) {
#pragma unroll
        for (int dy = 0; dy < DK; ++dy)
#pragma unroll
          for (int dx = 0; dx < DK; ++dx) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(halo + rbase + dy * PITCHF + dx * 16);
            const f32x4 w = yl_ld4(dwl + (dy * DK + dx) * p.Cin + cs);
            s.x = fmaf(v.x, w.x, s.x); s.y = fmaf(v.y, w.y, s.y);
            s.z = fmaf(v.z, w.z, s.z); s.w = fmaf(v.w, w.w, s.w);
          }
      } else {
#pragma unroll 1
        for (int dy = 0; dy < DK; ++dy) {
#pragma unroll
          for (int dx = 0; dx < DK; ++dx) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(halo + rbase + dy * PITCHF + dx * 16);
            const f32x4 w = yl_ld4(dwl + (dy * DK + dx) * p.Cin + cs);
            s.x = fmaf(v.x, w.x, s.x); s.y = fmaf(v.y, w.y, s.y);
            s.z = fmaf(v.z, w.z, s.z); s.w = fmaf(v.w, w.w, s.w);
          }
        }
      }
      // channel tail (c >= Cin): the packed 1x1 weights of those k slots are zero, no select needed
      bb[kb * 64 + lane] = yl_actc(s, p.dw_act, dlo, dhi);
      if (more || nxt) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");        // this block's tap reads are complete
        stage_store(stg);
      }
    }
    __syncthreads();                                                   // all B fragments of the tile are in LDS
    // ---- phase 2: this wave's n-tiles over all channel blocks, weights from registers
    if (p2) {
      const int b = tile / tiles_img;
      const int trem = tile - b * tiles_img;
      const int tyi = trem / tw, txi = trem - tyi * tw;
      YlPix px[1];
      px[0].b = b;
      px[0].oy = 4 * tyi + (pl >> 2);
      px[0].ox = 4 * txi + (pl & 3);
      px[0].valid = true;
      px[0].lin = ((size_t)b * p.OH + px[0].oy) * p.OW + px[0].ox;
      f32x4 acc[1][NTW];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int n = (nt0 + nt) * 16 + 4 * kq;
        acc[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (pre_add && n < p.N) acc[0][nt] = yl_ld4(p.res + px[0].lin * p.N + n);
      }
      f32x4 xn = bb[lane];
#pragma unroll
      for (int kb = 0; kb < KBMAX; ++kb) {
        if (kb < KB) {
          f32x4 xq[1];
          xq[0] = xn;
          if (kb + 1 < KBMAX && kb + 1 < KB) xn = bb[(kb + 1) * 64 + lane];
          yl_mma_step<NTW, 1>(wreg[kb], xq, acc);
        }
      }
      if (!pre_add && (p.res || p.up || p.act == YL_ACT_SILU)) yl_epi_generic<NTW, 1>(p, acc, px, nt0, kq);
      else yl_epi_fast<NTW, 1>(p, acc, px, nt0, kq, lo, hi, true);
    }
    if (dbuf) cur ^= 1;
    else __syncthreads();                                              // single buffer: reads done before the next tile's writes
  }
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
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kernel, 256, lds) != hipSuccess || nb < 1) nb = 1;
  if (nb > 4) nb = 4;
  cache[key] = nb * YL_NUM_CU;
  return nb * YL_NUM_CU;
}

template <int DK, int DS, int NTW>
hipError_t dwc_one(const YlConvMulti& m, int gx, size_t lds, int dbuf, hipStream_t st, bool attr_only, int* resident) {
  if (attr_only)
    return hipFuncSetAttribute((const void*)yl_conv_dwc_kernel<DK, DS, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               YL_DWC_LDS_MAX);
  if (resident) { *resident = yl_resident_blocks_c(yl_conv_dwc_kernel<DK, DS, NTW>, lds); return hipSuccess; }
  hipLaunchKernelGGL((yl_conv_dwc_kernel<DK, DS, NTW>), dim3(gx), dim3(256), lds, st, m, dbuf);
  return hipGetLastError();
}

template <int NTW>
hipError_t dwc_dk(const YlConvMulti& m, int dk, int ds, int gx, size_t lds, int dbuf, hipStream_t st, bool attr_only,
                  int* resident) {
  hipError_t e = hipSuccess;
  if (attr_only || (dk == 3 && ds == 1)) { e = dwc_one<3, 1, NTW>(m, gx, lds, dbuf, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || (dk == 3 && ds == 2)) { e = dwc_one<3, 2, NTW>(m, gx, lds, dbuf, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || (dk == 5 && ds == 1)) { e = dwc_one<5, 1, NTW>(m, gx, lds, dbuf, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || (dk == 5 && ds == 2)) { e = dwc_one<5, 2, NTW>(m, gx, lds, dbuf, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  return attr_only ? hipSuccess : hipErrorInvalidValue;
}

hipError_t dwc_any(const YlConvMulti& m, int ntw, int dk, int ds, int gx, size_t lds, int dbuf, hipStream_t st,
                   bool attr_only, int* resident) {
  hipError_t e = hipSuccess;
  if (attr_only || ntw == 1) { e = dwc_dk<1>(m, dk, ds, gx, lds, dbuf, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || ntw == 2) { e = dwc_dk<2>(m, dk, ds, gx, lds, dbuf, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || ntw == 3) { e = dwc_dk<3>(m, dk, ds, gx, lds, dbuf, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || ntw == 4) { e = dwc_dk<4>(m, dk, ds, gx, lds, dbuf, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  if (attr_only || ntw == 5) { e = dwc_dk<5>(m, dk, ds, gx, lds, dbuf, st, attr_only, resident); if (!attr_only || e != hipSuccess) return e; }
  return attr_only ? hipSuccess : hipErrorInvalidValue;
}

int kbmax_of(int ntw) { return ntw == 1 ? 18 : ntw == 2 ? 9 : ntw == 3 ? 6 : 4; }

}  // namespace

hipError_t yl_convc_init() {
  YlConvMulti m = {};
  return dwc_any(m, 0, 0, 0, 0, 0, 0, nullptr, true, nullptr);
}

// n problems of identical configuration (m.p[0..n-1] filled like for yl_conv_dwh_kernel).  Returns
// hipErrorNotSupported when the shape is outside this kernel's limits (the caller then takes the halo kernel).
hipError_t yl_launch_conv_dwc(YlConvMulti& m, hipStream_t st) {
  const YlConvP& p = m.p[0];
  const int ntw = (p.NTtot + 3) / 4;
  if (ntw < 1 || ntw > 5 || p.KB > kbmax_of(ntw) || (p.N & 3)) return hipErrorNotSupported;
  if (!((p.dw_k == 3 || p.dw_k == 5) && (p.dw_stride == 1 || p.dw_stride == 2))) return hipErrorNotSupported;
  const int HP = 3 * p.dw_stride + p.dw_k;
  const int PITCHF = ((HP * 16 + 7) / 64) * 64 + 56;
  const size_t fixed = ((((size_t)(p.dw_k * p.dw_k + 1) * p.Cin + 3) & ~(size_t)3) + (size_t)4 * HP * PITCHF) * 4;
  const size_t bb = (size_t)p.KB * 1024;
  // double-buffered B fragments (one barrier per tile) unless that costs a resident workgroup per CU
  int dbuf = 1;
  size_t lds = fixed + 2 * bb;
  const size_t budget = 160 * 1024;
  if (lds > YL_DWC_LDS_MAX || budget / lds < budget / (fixed + bb) ) { dbuf = 0; lds = fixed + bb; }
  if (lds > YL_DWC_LDS_MAX) return hipErrorNotSupported;
  long tiles[4], total = 0;
  for (int k = 0; k < m.n; ++k) {
    if ((m.p[k].OH & 3) || (m.p[k].OW & 3) || (size_t)m.p[k].B * m.p[k].H * m.p[k].W * m.p[k].Cin * 4 >= ((size_t)1 << 31))
      return hipErrorNotSupported;
    tiles[k] = (long)m.p[k].B * (m.p[k].OH >> 2) * (m.p[k].OW >> 2);
    total += tiles[k];
  }
  int res = 0;
  hipError_t e = dwc_any(m, ntw, p.dw_k, p.dw_stride, 0, lds, dbuf, st, false, &res);
  if (e != hipSuccess) return e;
  // every workgroup gets the same number of tiles (+-1): grid = tiles / rounds
  long gx = total;
  if (gx > res) {
    const long rounds = (total + res - 1) / res;
    gx = (total + rounds - 1) / rounds;
  }
  if (m.n == 1) { m.p[0].blk0 = 0; m.p[0].nblk = 0; }
  else {
    int at = 0;
    for (int k = 0; k < m.n; ++k) {
      long nb = (tiles[k] * gx + total / 2) / total;
      if (nb < 1) nb = 1;
      if (nb > tiles[k]) nb = tiles[k];
      m.p[k].blk0 = at;
      m.p[k].nblk = (int)nb;
      at += (int)nb;
    }
    gx = at;
  }
  return dwc_any(m, ntw, p.dw_k, p.dw_stride, (int)gx, lds, dbuf, st, false, nullptr);
}
