// Post-processing kernels for gfx950 (MI355X): anchor-free decode + score + threshold, and batched
// class-wise NMS (LDS bitonic sort + wave64 ballot suppression).
//
// Compiled with -ffp-contract=off: every expression below is evaluated with one rounding per
// operation in the association order of the reference expressions, so that results on identical
// inputs are identical to the reference's fp32 CPU arithmetic (up to libm ulp differences in
// expf/log1pf).
//
// Reference semantics implemented here (paths in the reference repository):
//   decode           scripts/helpers/utils_ms.py:71-106
//   score/threshold  tools/infer.py:463-475 (main), :310-340 (fallback), scripts/helpers/helpers.py:106-123
//   per-class NMS    tools/infer.py:476-493 + nms() :134-152 ; helpers.py:126-136
//   torchvision nms  stable sort by score desc; suppress j iff inter/(a_i+a_j-inter) > thr (NaN keeps)
//   greedy fallback  tools/infer.py:139-163  IoU = inter/(a1+a2-inter+1e-6); keep iff IoU <= thr
//   global top-k     tools/infer.py:368-379
//   back-map         tools/infer.py:508-516
#include "yl_internal.h"
#include <limits.h>
#include <math.h>

typedef unsigned long long u64;
typedef unsigned int u32;

#include "yl_decode.h"

// ------------------------------------------------------------------------------------------------
// decode + score: one lane per candidate.  A wave stages its 64 candidate rows (64*E contiguous
// floats inside one level) through a wave-private LDS region with coalesced reads, then every lane
// walks its own row (odd row pitch -> conflict-free ds_read_b32).
// grid = (ceil(N/128), B), block = 128 (2 waves).  HBM traffic: N*E*4 B read, N*24 B written / image.
template <bool STAGE>
__global__ __launch_bounds__(128) void yl_decode_score_kernel(YlLevels lv, int B, YlDecodeP p) {
  extern __shared__ __attribute__((aligned(16))) float yl_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int n0 = (blockIdx.x * 2 + wave) * 64;
  const int E = lv.E, Ep = E | 1;
  float* rows = yl_smem + wave * 64 * Ep;
  if (STAGE) {
    const int l0 = yl_level_of(lv, n0);
    const int nend = (n0 + 64 < lv.N) ? n0 + 64 : lv.N;
    if (n0 < lv.N && nend <= lv.off[l0 + 1]) {
      // fast path: the wave's rows are one contiguous run of (nend-n0)*E floats inside level l0.
      // Flat coalesced copy (256 B per wave instruction, 8 loads in flight), scattered into the
      // odd-pitch row layout; (row, col) of element lane+64*j is tracked incrementally.
      const int nl = lv.A[l0] * lv.S[l0] * lv.S[l0];
      const float* src = lv.ptr[l0] + ((size_t)b * nl + (n0 - lv.off[l0])) * E;
      const int total = (nend - n0) * E;
      const int qs = 64 / E, rs = 64 - qs * E;           // per-iteration (row, col) increments
      int row = lane / E, col = lane - row * E;
      for (int i0 = 0; i0 < total; i0 += 64 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * 64 + lane;
          v[u] = (i < total) ? src[i] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * 64 + lane;
          if (i < total) rows[row * Ep + col] = v[u];
          row += qs; col += rs;
          if (col >= E) { col -= E; ++row; }
        }
      }
    } else {
      // generic path (level boundary inside the wave's run): row by row
      for (int r = 0; r < 64; ++r) {
        const int n = n0 + r;
        if (n >= lv.N) break;
        const int l = yl_level_of(lv, n);
        const int nl = lv.A[l] * lv.S[l] * lv.S[l];
        const float* src = lv.ptr[l] + ((size_t)b * nl + (n - lv.off[l])) * E;
        for (int c = lane; c < E; c += 64) rows[r * Ep + c] = src[c];
      }
    }
    __syncthreads();
  }
  const int n = n0 + lane;
  if (n >= lv.N) return;
  const int l = yl_level_of(lv, n);
  const int r = n - lv.off[l];
  const float* row;
  if (STAGE) {
    row = rows + lane * Ep;
  } else {
    const int nl = lv.A[l] * lv.S[l] * lv.S[l];
    row = lv.ptr[l] + ((size_t)b * nl + r) * E;
  }
  float px, py, pw, ph;
  yl_decode_box(lv, l, r, row[0], row[1], row[2], row[3], p.center_mode, p.wh_mode, px, py, pw, ph);
  const float obj = yl_sigmoid(row[4]);
  float score;
  int ci = 0;
  const int C = lv.C;
  if (C > 1) {
    // reference: (conf, idx) = sigmoid(cls).max(-1), first maximum wins.  sigmoid is monotonic, so
    // conf = sigmoid(max logit); the index can only differ from the logit arg-max when an EARLIER class
    // has a (slightly) smaller logit whose sigmoid rounds to the same float (saturation / <1 ulp):
    // only those candidates get their own sigmoid.  1-2 expf per candidate instead of C.
    float lmax = row[5];
    for (int c = 1; c < C; ++c) {
      const float l = row[5 + c];
      if (l > lmax) { lmax = l; ci = c; }
    }
    const float best = yl_sigmoid(lmax);
    // logits whose sigmoid can round to `best`: within 1e-3 relative of lmax; anything above 10 when the
    // result is near saturation (float spacing below 1.0 is 6e-8 = e^-16.6); everything when `best` is
    // subnormal or zero (coarse spacing / underflow of 1/(1+inf))
    const float band = lmax - 1e-3f * (1.0f + fabsf(lmax));
    const bool wide = best < 1.2e-38f;
    for (int c = 0; c < ci; ++c) {
      const float l = row[5 + c];
      if ((wide || l >= band || l > 10.0f) && yl_sigmoid(l) == best) { ci = c; break; }
    }
    score = obj * best;
  } else if (C == 1 && p.mode == YL_POST_FALLBACK) {
    score = obj * yl_sigmoid(row[5]);             // tools/infer.py:316-320
  } else {
    score = obj;                                  // tools/infer.py:470-472, helpers.py:113-115
  }
  if (p.mode == YL_POST_FALLBACK && !(pw >= 2.0f && ph >= 2.0f)) score = -INFINITY;  // :334-340
  float4 bx;
  bx.x = yl_clampf(px - pw * 0.5f, 0.0f, lv.hi);
  bx.y = yl_clampf(py - ph * 0.5f, 0.0f, lv.hi);
  bx.z = yl_clampf(px + pw * 0.5f, 0.0f, lv.hi);
  bx.w = yl_clampf(py + ph * 0.5f, 0.0f, lv.hi);
  const size_t o = (size_t)b * lv.N + n;
  p.boxes[o] = bx;
  p.scores[o] = score;
  p.cls[o] = ci;
}

// decode only (decode_preds_anchorfree): box + obj logits; class logits copied by a second kernel.
__global__ __launch_bounds__(256) void yl_decode_box_kernel(YlLevels lv, int B, int center_mode, int wh_mode,
                                                           float4* box, float* obj) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= lv.N) return;
  const int l = yl_level_of(lv, n);
  const int r = n - lv.off[l];
  const int nl = lv.A[l] * lv.S[l] * lv.S[l];
  const float* row = lv.ptr[l] + ((size_t)b * nl + r) * lv.E;
  float px, py, pw, ph;
  yl_decode_box(lv, l, r, row[0], row[1], row[2], row[3], center_mode, wh_mode, px, py, pw, ph);
  float4 bx;
  bx.x = yl_clampf(px - pw * 0.5f, 0.0f, lv.hi);
  bx.y = yl_clampf(py - ph * 0.5f, 0.0f, lv.hi);
  bx.z = yl_clampf(px + pw * 0.5f, 0.0f, lv.hi);
  bx.w = yl_clampf(py + ph * 0.5f, 0.0f, lv.hi);
  box[(size_t)b * lv.N + n] = bx;
  obj[(size_t)b * lv.N + n] = row[4];
}

__global__ __launch_bounds__(256) void yl_copy_cls_kernel(YlLevels lv, int B, float* cls) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // over N*C of one image
  const int b = blockIdx.y;
  const int C = lv.C;
  if (i >= (size_t)lv.N * C) return;
  const int n = (int)(i / C), c = (int)(i % C);
  const int l = yl_level_of(lv, n);
  const int nl = lv.A[l] * lv.S[l] * lv.S[l];
  cls[(size_t)b * lv.N * C + i] = lv.ptr[l][((size_t)b * nl + (n - lv.off[l])) * lv.E + 5 + c];
}

// ------------------------------------------------------------------------------------------------
// NMS.  One 1024-thread workgroup per image.
//   key = class(12) | ~orderable(score)(32) | candidate index(20): an ascending sort yields
//   class asc, score desc, index asc == per-class stable descending sort of the masked candidates.
__device__ __forceinline__ u32 yl_desc_bits(float s) {
  u32 u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-orderable
  return ~u;                                        // descending
}

// exact reference predicate (one rounding per operation, IEEE division)
__device__ __forceinline__ bool yl_suppress_exact(float inter, float ai, float aj, float thr, int impl) {
  if (impl == YL_NMS_TORCHVISION) {
    const float ovr = inter / (ai + aj - inter);
    return ovr > thr;                               // NaN (0/0) never suppresses
  }
  const float iou = inter / (ai + aj - inter + 1e-6f);
  return !(iou <= thr);                             // reference keeps iff iou <= thr
}

__device__ __forceinline__ float yl_inter(const float4& bi, const float4& bj) {
  const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
  const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
  const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
  return w * h;
}

// Same result as yl_suppress_exact for every input, but the IEEE division (~12 dependent VALU ops)
// only runs when some relevant lane of the wave is within 1e-5 (relative) of the threshold: outside
// that band  inter <> thr*den*(1 +- 1e-5)  decides identically (division / product rounding is 6e-8).
__device__ __forceinline__ bool yl_suppress(const float4& bi, float ai, const float4& bj, float aj, float thr,
                                            int impl, bool relevant) {
  const float inter = yl_inter(bi, bj);
  const float den = (impl == YL_NMS_TORCHVISION) ? (ai + aj - inter) : (ai + aj - inter + 1e-6f);
  const float cmp = thr * den;
  const bool sure_yes = den > 0.0f && inter > cmp * 1.00001f;
  const bool sure_no = den > 0.0f && inter < cmp * 0.99999f && thr >= 0.0f;
  bool r = sure_yes;
  if (__ballot(relevant && !(sure_yes || sure_no)) != 0ull) r = yl_suppress_exact(inter, ai, aj, thr, impl);
  return r;
}

__device__ __forceinline__ float yl_readlane_f(float v, int lane_uniform) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
}

__device__ __forceinline__ float4 yl_shfl4(const float4& v, int src) {
  float4 r;
  r.x = __shfl(v.x, src); r.y = __shfl(v.y, src); r.z = __shfl(v.z, src); r.w = __shfl(v.w, src);
  return r;
}

#ifdef YL_NMS_STAMP
// profiling aid (variant builds only, tools/build_variant.sh ... -DYL_NMS_STAMP): wall-clock ticks (100 MHz) at the
// phase boundaries of block 0
__device__ unsigned long long yl_nms_stamps[64];     // [workgroup of image 0 (blockIdx.y)][16]
__device__ unsigned long long yl_nms_tstamps[128];
__device__ int yl_nms_tcount;
#define YL_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) yl_nms_stamps[(blockIdx.y & 3) * 16 + (i)] = wall_clock64(); } while (0)
extern "C" int yl_debug_nms_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(yl_nms_stamps), sizeof(unsigned long long) * 64);
}
extern "C" int yl_debug_nms_tstamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(yl_nms_tstamps), sizeof(unsigned long long) * 128);
}
#define YL_TSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 128 && yl_nms_tstamps[i] == 0) yl_nms_tstamps[i] = wall_clock64(); } while (0)
#else
#define YL_STAMP(i) do {} while (0)
#define YL_TSTAMP(i) do {} while (0)
#endif
// Scalar pass of the in-tile resolution: candidate i (alive mask `am`, ascending = score order) is kept iff
// none of its suppressors (per-lane mask M, bits < lane only) is kept.  Candidates nobody can suppress
// (M == 0) are kept unconditionally, so only the "suspects" are walked -- usually a handful.
__device__ __forceinline__ u64 yl_tile_scan(u64 am, u64 M) {
  const u64 sus = __ballot(M != 0ull) & am;
  u64 kept = am & ~sus;
  const u32 mlo = (u32)M, mhi = (u32)(M >> 32);
  for (u64 rem = sus; rem; rem &= rem - 1) {
    const int i = __ffsll((long long)rem) - 1;
    const u64 mi = ((u64)(u32)__builtin_amdgcn_readlane((int)mhi, i) << 32) |
                   (u64)(u32)__builtin_amdgcn_readlane((int)mlo, i);
    if ((mi & kept) == 0ull) kept |= 1ull << i;
  }
  return kept;
}

// Greedy resolution INSIDE one 64-candidate tile (lane = candidate in score order, `alive` = not yet
// suppressed by boxes kept earlier).  Pass 1 (vector, no scalar round trips): for every alive candidate i,
// broadcast its box through SGPRs and let every later lane j record "i would suppress me" in a 64-bit
// per-lane mask M.  Pass 2 (scalar): walk the alive candidates in order; i is kept iff none of its
// suppressors is kept: (M_i & kept) == 0.  Identical keep set to the serial "take first alive, strike the
// rest" loop it replaces, whose every step paid a VALU->SALU->VALU round trip (~770 clk per kept box).
// Pairs inside the 1e-5 band around the threshold are re-decided with the exact IEEE division.
__device__ __forceinline__ bool yl_tile_resolve(const float4& bj, float aj, bool alive, float thr, int impl, int lane) {
  const u64 am = __ballot(alive);
  u64 M = 0ull, U = 0ull;
  for (u64 rem = am; rem; rem &= rem - 1) {
    const int i = __ffsll((long long)rem) - 1;
    const u64 bit = 1ull << i;
    const float4 bt = make_float4(yl_readlane_f(bj.x, i), yl_readlane_f(bj.y, i), yl_readlane_f(bj.z, i),
                                  yl_readlane_f(bj.w, i));
    const float at = yl_readlane_f(aj, i);
    const float inter = yl_inter(bt, bj);
    const float den = (impl == YL_NMS_TORCHVISION) ? (at + aj - inter) : (at + aj - inter + 1e-6f);
    const float cmp = thr * den;
    const bool sure_yes = den > 0.0f && inter > cmp * 1.00001f;
    const bool sure_no = den > 0.0f && inter < cmp * 0.99999f && thr >= 0.0f;
    const bool rel = lane > i;
    if (rel && sure_yes) M |= bit;
    if (rel && !(sure_yes || sure_no)) U |= bit;
  }
  if (__ballot(U != 0ull) != 0ull) {                          // rare: exact predicate for the undecided pairs
    for (u64 rem = U; rem; rem &= rem - 1) {
      const int i = __ffsll((long long)rem) - 1;
      const float4 bt = yl_shfl4(bj, i);
      const float at = __shfl(aj, i);
      if (yl_suppress_exact(yl_inter(bt, bj), at, aj, thr, impl)) M |= 1ull << i;
    }
  }
  const u64 kept = yl_tile_scan(am, M);
  return ((kept >> lane) & 1ull) != 0ull;
}

// class segments above YL_NMS_BIG survivors are handled by the whole workgroup (yl_nms_segment_block)
#ifndef YL_NMS_BIG
#define YL_NMS_BIG 64
#endif
#define YL_NMS_BIGQ 200
#define YL_NMS_SCRATCH 2560          // bytes of LDS behind the keys: ints [0..3] counters, [8..209] big-class queue,
                                     // byte 1024..2063: 2 x 65 u64 OR scratch of the cooperative pass,
                                     // byte 2304..2559: class -> workgroup owner table (class-group split, C <= 256)

// Bitonic sort of P = KPT * 1024 keys held in LDS by a 1024-thread workgroup, KPT keys per thread in
// REGISTERS (thread t owns positions KPT*t .. KPT*t+KPT-1).  Of the log2(P)*(log2(P)+1)/2 compare-exchange
// stages only those whose partner distance j reaches another wave (j >= 64*KPT) go through LDS with a
// workgroup barrier; 64*KPT > j >= KPT exchange with lane ^ (j/KPT) by wave shuffles and j < KPT stays inside
// the thread.  P = 8192: 10 barrier stages instead of 91 (measured 65 -> 16 us per image on the benchmark's
// 4335-survivor images).  Same network, same result as yl_bitonic_sort.
template <int KPT>
__device__ __forceinline__ void yl_bitonic_sort_reg(u64* keys, int tid) {
  constexpr int P = KPT * 1024;
  const int lane = tid & 63;
  const int base = KPT * tid;
  u64 v[KPT];
#pragma unroll
  for (int r = 0; r < KPT; ++r) v[r] = keys[base + r];
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    int j = k >> 1;
    if (j >= KPT * 64) {                                   // partner in another wave: through LDS
#pragma unroll
      for (int r = 0; r < KPT; ++r) keys[base + r] = v[r];
      __syncthreads();
      for (; j >= KPT * 64; j >>= 1) {
        for (int i = tid; i < (P >> 1); i += 1024) {
          const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
          const int hi = lo | j;
          const u64 a = keys[lo], c = keys[hi];
          const bool up = (lo & k) == 0;
          if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
        }
        __syncthreads();
      }
#pragma unroll
      for (int r = 0; r < KPT; ++r) v[r] = keys[base + r];
      __syncthreads();
    }
    for (; j >= KPT; j >>= 1) {                            // partner in the same wave: lane ^ (j / KPT)
      const int m = j / KPT;
      const bool lower = (lane & m) == 0;
#pragma unroll
      for (int r = 0; r < KPT; ++r) {
        const u64 o = __shfl_xor(v[r], m, 64);
        const bool up = ((base + r) & k) == 0;
        const u64 mn = v[r] < o ? v[r] : o, mx = v[r] < o ? o : v[r];
        v[r] = (lower == up) ? mn : mx;
      }
    }
#pragma unroll
    for (int jj = KPT >> 1; jj >= 1; jj >>= 1) {           // partner in the same thread
      if (jj > j) continue;
#pragma unroll
      for (int r = 0; r < KPT; ++r) {
        if (r & jj) continue;
        const u64 a = v[r], c = v[r | jj];
        const bool up = ((base + r) & k) == 0;
        if ((a > c) == up) { v[r] = c; v[r | jj] = a; }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < KPT; ++r) keys[base + r] = v[r];
  __syncthreads();
}

template <typename KeyPtr>
__device__ __forceinline__ void yl_bitonic_sort(KeyPtr keys, int P, int tid, int nthreads) {
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < (P >> 1); i += nthreads) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int hi = lo | j;
        const u64 a = keys[lo], c = keys[hi];
        const bool up = (lo & k) == 0;
        if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
}

// greedy NMS of one class segment [s,e) of the sorted key array by ONE wave.  SBOX: the survivors' boxes
// were gathered into LDS in sorted order (sbox[pos]); every box access in the loops below is then an LDS
// read -- the kept-box loop would otherwise be a chain of dependent global gathers (~1 us each).
// k32 views the 64-bit keys as u32 pairs: [2*pos] = candidate index, [2*pos+1] = kept list of the
// segment (entry s+k holds the sorted position of the k-th kept box).
template <bool SBOX, typename K32Ptr>
__device__ __forceinline__ int yl_nms_segment(K32Ptr k32, int s, int e, const float4* __restrict__ boxes,
                                              const float4* sbox, float thr, int impl, int cap, int lane) {
  int nk = 0;
  for (int cs = s; cs < e && nk < cap; cs += 64) {
    const int pos = cs + lane;
    const bool valid = pos < e;
    float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) bj = SBOX ? sbox[pos] : boxes[k32[2 * pos]];
    const float aj = (bj.z - bj.x) * (bj.w - bj.y);
    bool alive = valid;
    // boxes kept by earlier chunks: fetched 64 at a time (lane k holds kept box k0+k: two LDS reads, all in
    // flight together) and broadcast through SGPRs -- no dependent LDS read in the per-box loop
    for (int k0 = 0; k0 < nk && __ballot(alive) != 0ull; k0 += 64) {
      const int kk = (nk - k0) < 64 ? (nk - k0) : 64;
      float4 kb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < kk) {
        const u32 q = k32[2 * (s + k0 + lane) + 1];
        kb = SBOX ? sbox[q] : boxes[k32[2 * q]];
      }
      const float ka = (kb.z - kb.x) * (kb.w - kb.y);
      for (int k = 0; k < kk; ++k) {
        const int ku = __builtin_amdgcn_readfirstlane(k);
        const float4 bi = make_float4(yl_readlane_f(kb.x, ku), yl_readlane_f(kb.y, ku), yl_readlane_f(kb.z, ku),
                                      yl_readlane_f(kb.w, ku));
        const float ai = yl_readlane_f(ka, ku);
        if (yl_suppress(bi, ai, bj, aj, thr, impl, alive) && alive) alive = false;
      }
    }
    alive = yl_tile_resolve(bj, aj, alive, thr, impl, lane);
    const u64 mask = __ballot(alive);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (alive && nk + rank < cap) k32[2 * (s + nk + rank) + 1] = (u32)pos;
    nk += __popcll(mask);
  }
  return nk < cap ? nk : cap;
}

// greedy NMS of ONE LARGE class segment [s,e) by the WHOLE workgroup.  Every pair test costs a wave ~400
// clocks (a ~35-deep dependent VALU chain behind an SGPR broadcast; one wave per SIMD, nothing to overlap
// with), and the per-wave routine above runs nk + 64 of them back to back for every 64-candidate chunk:
// measured 130 of 160 us per image on the B=64 benchmark, spent by ONE wave on a 1178-survivor class while
// 15 waves idled.  Here all waves work on the same chunk (lane = candidate) and split the SOURCE boxes:
//   phase A  kept boxes k = wave, wave+nwaves, ...  -> "dead" ballots OR-ed in LDS        (nk/nwaves tests)
//   phase B  in-chunk suppressor rows i = wave*(64/nwaves).. -> partial masks M_j OR-ed in LDS (4 tests)
//   then every wave runs the same scalar scan (kept iff (M_i & kept) == 0) -- no result to publish.
// Two barriers per chunk; the LDS scratch is double-buffered by chunk parity.  Same keep set and order
// as yl_nms_segment.  s_or: 2 x (1 + 64) u64 of LDS, zero on entry and on exit.
template <bool SBOX, typename K32Ptr>
__device__ __forceinline__ int yl_nms_segment_block(K32Ptr k32, int s, int e, const float4* __restrict__ boxes,
                                                    const float4* sbox, float thr, int impl, int cap, u64* s_or) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int rows = (64 + nwaves - 1) / nwaves;              // phase-B rows per wave
  int nk = 0, par = 0;
  for (int cs = s; cs < e && nk < cap; cs += 64, par ^= 1) {
    const int ci = (cs - s) >> 6;
    YL_TSTAMP(6 * ci);
    u64* s_dead = s_or + par * 65;
    u64* s_M = s_dead + 1;
    const int pos = cs + lane;
    const bool valid = pos < e;
    float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) bj = SBOX ? sbox[pos] : boxes[k32[2 * pos]];
    const float aj = (bj.z - bj.x) * (bj.w - bj.y);
    // ---- phase A: my share of the kept boxes
    bool alive = valid;
    for (int k0 = wave; k0 < nk && __ballot(alive) != 0ull; k0 += 64 * nwaves) {
      const int kidx = k0 + lane * nwaves;
      float4 kb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kidx < nk) {
        const u32 q = k32[2 * (s + kidx) + 1];
        kb = SBOX ? sbox[q] : boxes[k32[2 * q]];
      }
      const float ka = (kb.z - kb.x) * (kb.w - kb.y);
      const int kk = (nk - k0 + nwaves - 1) / nwaves;       // kept boxes in this batch (<= 64)
      for (int k = 0; k < kk && k < 64; ++k) {
        const int ku = __builtin_amdgcn_readfirstlane(k);
        const float4 bi = make_float4(yl_readlane_f(kb.x, ku), yl_readlane_f(kb.y, ku), yl_readlane_f(kb.z, ku),
                                      yl_readlane_f(kb.w, ku));
        const float ai = yl_readlane_f(ka, ku);
        if (yl_suppress(bi, ai, bj, aj, thr, impl, alive) && alive) alive = false;
      }
    }
    {
      const u64 dm = __ballot(valid && !alive);
      if (lane == 0 && dm) atomicOr((unsigned long long*)s_dead, (unsigned long long)dm);
    }
    YL_TSTAMP(6 * ci + 1);
    __syncthreads();
    YL_TSTAMP(6 * ci + 2);
    if (wave == 0) {                                        // zero the other parity's scratch for the next chunk
      if (lane == 0) s_or[(par ^ 1) * 65] = 0ull;
      s_or[(par ^ 1) * 65 + 1 + lane] = 0ull;
    }
    alive = valid && ((s_dead[0] >> lane) & 1ull) == 0ull;
    const u64 am = __ballot(alive);
    // ---- phase B: my rows of the in-chunk suppression matrix
    {
      u64 M = 0ull, U = 0ull;
      for (int r = 0; r < rows; ++r) {
        const int i = __builtin_amdgcn_readfirstlane(wave * rows + r);
        if (i >= 64 || ((am >> i) & 1ull) == 0ull) continue;
        const u64 bit = 1ull << i;
        const float4 bt = make_float4(yl_readlane_f(bj.x, i), yl_readlane_f(bj.y, i), yl_readlane_f(bj.z, i),
                                      yl_readlane_f(bj.w, i));
        const float at = yl_readlane_f(aj, i);
        const float inter = yl_inter(bt, bj);
        const float den = (impl == YL_NMS_TORCHVISION) ? (at + aj - inter) : (at + aj - inter + 1e-6f);
        const float cmp = thr * den;
        const bool sure_yes = den > 0.0f && inter > cmp * 1.00001f;
        const bool sure_no = den > 0.0f && inter < cmp * 0.99999f && thr >= 0.0f;
        const bool rel = lane > i;
        if (rel && sure_yes) M |= bit;
        if (rel && !(sure_yes || sure_no)) U |= bit;
      }
      if (__ballot(U != 0ull) != 0ull) {
        for (u64 rem = U; rem; rem &= rem - 1) {
          const int i = __ffsll((long long)rem) - 1;
          const float4 bt = yl_shfl4(bj, i);
          const float at = __shfl(aj, i);
          if (yl_suppress_exact(yl_inter(bt, bj), at, aj, thr, impl)) M |= 1ull << i;
        }
      }
      if (M) atomicOr((unsigned long long*)&s_M[lane], (unsigned long long)M);
    }
    YL_TSTAMP(6 * ci + 3);
    __syncthreads();
    YL_TSTAMP(6 * ci + 4);
    // ---- scan (every wave, identical result)
    const u64 kept = yl_tile_scan(am, s_M[lane]);
    {
      // kept entry `slot` is re-read in phase A by wave slot % nwaves only: that wave writes it (program order
      // within a wave, no barrier needed before the next chunk)
      const int slot = nk + __popcll(kept & ((1ull << lane) - 1ull));
      if (((kept >> lane) & 1ull) && slot < cap && (slot % nwaves) == wave) k32[2 * (s + slot) + 1] = (u32)pos;
    }
    nk += __popcll(kept);
    YL_TSTAMP(6 * ci + 5);
  }
  __syncthreads();                                          // kept list complete; scratch of the last parity:
  if (wave == 0) {
    if (lane == 0) { s_or[0] = 0ull; s_or[65] = 0ull; }
    s_or[1 + lane] = 0ull; s_or[66 + lane] = 0ull;
  }
  __syncthreads();
  return nk < cap ? nk : cap;
}

__device__ __forceinline__ void yl_write_det(const YlNmsP& p, int b, float* dst, int* dst_idx, int orow,
                                             const float4& bx, float score, int c, int idx, bool mapped) {
  float x1 = bx.x, y1 = bx.y, x2 = bx.z, y2 = bx.w;
  if (mapped && p.backmap) {                                 // tools/infer.py:508-516
    const float* m = p.backmap + b * 5;
    x1 = (x1 - m[0]) / m[2]; x2 = (x2 - m[0]) / m[2];
    y1 = (y1 - m[1]) / m[2]; y2 = (y2 - m[1]) / m[2];
    x1 = yl_clampf(x1, 0.f, m[3] - 1.f); x2 = yl_clampf(x2, 0.f, m[3] - 1.f);
    y1 = yl_clampf(y1, 0.f, m[4] - 1.f); y2 = yl_clampf(y2, 0.f, m[4] - 1.f);
  }
  float* d = dst + (size_t)orow * 6;
  d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2; d[4] = score; d[5] = (float)c;
  if (dst_idx) dst_idx[orow] = idx;
}

// Body of the NMS kernel, instantiated once for LDS key storage and once for the global-memory
// fallback so that each gets address-space-specific code after inlining.
template <bool LDS_KEYS, bool SBOX>
__device__ __forceinline__ void yl_nms_run(const YlNmsP& p, u64* keys, int P, int nsurv, int b, int* s_misc,
                                           float4* sbox, const unsigned char* owner /*LDS [C], nullptr if G == 1*/) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int N = p.N, C = p.C;
  const float4* boxes = p.boxes + (size_t)b * N;
  const float* scores = p.scores + (size_t)b * N;
  const int* cls = p.cls ? p.cls + (size_t)b * N : nullptr;
  int* ws_start = p.cls_ws + (size_t)b * 4 * C;
  int* ws_end = ws_start + C;
  int* ws_kept = ws_end + C;
  int* ws_off = ws_kept + C;
  u32* k32 = (u32*)keys;
  const int G = p.G > 1 ? p.G : 1, g = G > 1 ? (int)blockIdx.y : 0;     // my class group: owner[c] == g

  YL_STAMP(1);
  if (tid == 0) s_misc[1] = 0;
  for (int c = tid; c < C; c += blockDim.x)
    if (G == 1 || owner[c] == g) { ws_start[c] = -1; ws_end[c] = 0; ws_kept[c] = 0; ws_off[c] = 0; }
  __syncthreads();
  for (int n0 = 0; n0 < N; n0 += blockDim.x) {                // wave-aggregated slot allocation (order is irrelevant:
    const int n = n0 + tid;                                    // the keys are sorted next)
    const float sc = n < N ? scores[n] : -INFINITY;
    const bool act = n < N && sc > p.conf_thr && (G == 1 || owner[cls[n]] == g);
    const u64 m = __ballot(act);
    if (m == 0ull) continue;
    int base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(&s_misc[1], __popcll(m));
    base = __builtin_amdgcn_readlane(base, leader);
    if (act) {
      const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
      const u64 c = cls ? (u64)cls[n] : 0ull;
      keys[slot] = (c << 52) | ((u64)yl_desc_bits(sc) << 20) | (u64)n;
    }
  }
  for (int i = nsurv + tid; i < P; i += blockDim.x) keys[i] = ~0ull;
  __syncthreads();
  YL_STAMP(2);
  if (LDS_KEYS && blockDim.x == 1024 && P >= 1024) {
    u64* lk = (u64*)keys;
    switch (P >> 10) {
      case 1: yl_bitonic_sort_reg<1>(lk, tid); break;
      case 2: yl_bitonic_sort_reg<2>(lk, tid); break;
      case 4: yl_bitonic_sort_reg<4>(lk, tid); break;
      case 8: yl_bitonic_sort_reg<8>(lk, tid); break;
      default: yl_bitonic_sort_reg<16>(lk, tid); break;
    }
  } else {
    yl_bitonic_sort(keys, P, tid, blockDim.x);
  }
  YL_STAMP(3);

  for (int pos = tid; pos < nsurv; pos += blockDim.x) {
    const int c = (int)(keys[pos] >> 52);
    if (pos == 0 || (int)(keys[pos - 1] >> 52) != c) ws_start[c] = pos;
    if (pos == nsurv - 1 || (int)(keys[pos + 1] >> 52) != c) ws_end[c] = pos + 1;
  }
  __syncthreads();
  for (int pos = tid; pos < nsurv; pos += blockDim.x) {
    const u64 idx = keys[pos] & 0xFFFFFull;
    keys[pos] = idx;
    if (SBOX) sbox[pos] = boxes[idx];                      // all gathers in flight at once
  }
  __syncthreads();
  YL_STAMP(4);

  // small classes: one wave per class; large classes (> YL_NMS_BIG survivors) are queued for the cooperative pass
  int* s_big = s_misc + 8;                                    // [0] count, [1..] class ids (YL_NMS_BIGQ entries)
  if (tid == 0) s_big[0] = 0;
  if (tid < 130) reinterpret_cast<u64*>(s_misc + 256)[tid] = 0ull;   // OR scratch of the cooperative pass
  __syncthreads();
  for (int c = wave; c < C; c += nwaves) {
    if (G > 1 && owner[c] != g) continue;
    const int s = ws_start[c];
    if (s < 0) continue;
    const int e = ws_end[c];
    if (e - s > YL_NMS_BIG && e - s <= 64 * 32 * nwaves) {
      int slot = YL_NMS_BIGQ;
      if (lane == 0) slot = atomicAdd(&s_big[0], 1);
      slot = __builtin_amdgcn_readfirstlane(slot);
      if (slot < YL_NMS_BIGQ) {
        if (lane == 0) s_big[1 + slot] = c;
        continue;
      }
    }
    const int nk = yl_nms_segment<SBOX>(k32, s, e, boxes, sbox, p.iou_thr, p.impl, p.cap, lane);
    if (lane == 0) ws_kept[c] = nk;
  }
  __syncthreads();
  {
    const int nbig = s_big[0] < YL_NMS_BIGQ ? s_big[0] : YL_NMS_BIGQ;
    for (int q = 0; q < nbig; ++q) {
      const int c = s_big[1 + q];
      const int nk = yl_nms_segment_block<SBOX>(k32, ws_start[c], ws_end[c], boxes, sbox, p.iou_thr, p.impl, p.cap,
                                                reinterpret_cast<u64*>(s_misc + 256));
      if (tid == 0) ws_kept[c] = nk;
    }
  }
  __syncthreads();
  YL_STAMP(5);

  if (G > 1) {
    // ---- class-group split: publish my classes' kept candidates (per-class start / count are already in cls_ws)
    int* klist = p.kept_list + ((size_t)b * G + g) * N;
    for (int c = wave; c < C; c += nwaves) {
      if (owner[c] != g) continue;
      const int s = ws_start[c];
      if (s < 0) continue;
      const int nk = ws_kept[c];
      for (int k = lane; k < nk; k += 64) klist[s + k] = (int)k32[2 * k32[2 * (s + k) + 1]];
    }
    // group 0 also leaves the class -> group table for the merge kernel (yl_nms_merge_kernel, next in the stream:
    // the kernel boundary orders these plain stores, no device-scope fences or arrival counters needed)
    if (g == 0) {
      unsigned char* ow = reinterpret_cast<unsigned char*>(p.done) + (size_t)b * 256;
      for (int c = tid; c < C; c += blockDim.x) ow[c] = owner[c];
    }
    YL_STAMP(6);
#ifdef YL_NMS_STAMP
    if (blockIdx.x == 0 && tid == 0) yl_nms_stamps[(blockIdx.y & 3) * 16 + 7] = (unsigned long long)nsurv;
#endif
    return;
  }

  // -- output offsets: exclusive scan of the kept counts over classes (wave 0)
  if (wave == 0) {
    int running = 0;
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int c = c0 + lane;
      const int v = (c < C) ? ws_kept[c] : 0;
      int incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      if (c < C) ws_off[c] = running + incl - v;
      running += __shfl(incl, 63);
    }
    if (lane == 0) s_misc[2] = running;
  }
  __syncthreads();
  const int total = s_misc[2];
  const bool do_topk = (p.topk > 0) && (total > p.topk);
  float* dst = do_topk ? p.tmp_dets + (size_t)b * N * 6 : p.dets + (size_t)b * p.max_out * 6;
  int* dst_idx = do_topk ? p.tmp_idx + (size_t)b * N : (p.keep_idx ? p.keep_idx + (size_t)b * p.max_out : nullptr);
  const int dst_rows = do_topk ? N : p.max_out;
  for (int c = wave; c < C; c += nwaves) {
    const int s = ws_start[c];
    if (s < 0) continue;
    const int nk = ws_kept[c], off = ws_off[c];
    for (int k = lane; k < nk; k += 64) {
      const int orow = off + k;
      if (orow >= dst_rows) continue;
      const u32 pos = k32[2 * (s + k) + 1];
      const u32 idx = k32[2 * pos];
      yl_write_det(p, b, dst, dst_idx, orow, SBOX ? sbox[pos] : boxes[idx], scores[idx], c, (int)idx, !do_topk);
    }
  }
  if (!do_topk) {
    YL_STAMP(6);
    if (tid == 0) { p.counts[b] = total; }
#ifdef YL_NMS_STAMP
    if (blockIdx.x == 0 && tid == 0) { yl_nms_stamps[7] = (unsigned long long)nsurv; yl_nms_stamps[8] = (unsigned long long)total; }
#endif
    return;
  }
  // -- fallback global top-k (tools/infer.py:377-379): sort kept rows by score desc, take topk
  __syncthreads();
  int P2 = 64;
  while (P2 < total) P2 <<= 1;
  for (int i = tid; i < P2; i += blockDim.x)
    keys[i] = (i < total) ? (((u64)yl_desc_bits(dst[(size_t)i * 6 + 4]) << 32) | (u64)i) : ~0ull;
  __syncthreads();
  yl_bitonic_sort(keys, P2, tid, blockDim.x);
  float* fin = p.dets + (size_t)b * p.max_out * 6;
  int* fin_idx = p.keep_idx ? p.keep_idx + (size_t)b * p.max_out : nullptr;
  for (int r = tid; r < p.topk && r < p.max_out; r += blockDim.x) {
    const u32 i = (u32)(keys[r] & 0xFFFFFFFFull);
    const float* srow = dst + (size_t)i * 6;
    const float4 bx = make_float4(srow[0], srow[1], srow[2], srow[3]);
    yl_write_det(p, b, fin, fin_idx, r, bx, srow[4], (int)srow[5], dst_idx[i], true);
  }
  if (tid == 0) p.counts[b] = p.topk;
}

__global__ __launch_bounds__(1024) void yl_nms_kernel(YlNmsP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char yl_smem_raw[];
  // dynamic LDS: [lds_cap] u64 keys, then YL_NMS_SCRATCH bytes: 4 counters, the big-class queue, the OR scratch
  u64* lkeys = (u64*)yl_smem_raw;
  int* s_misc = (int*)(yl_smem_raw + (size_t)p.lds_cap * 8);
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* scores = p.scores + (size_t)b * p.N;
  YL_STAMP(0);
  if (tid == 0) s_misc[0] = 0;
  __syncthreads();
  const int G = p.G > 1 ? p.G : 1, g = G > 1 ? (int)blockIdx.y : 0;
  const int* clsb = p.cls ? p.cls + (size_t)b * p.N : nullptr;
  unsigned char* owner = nullptr;
  int nsurv;
  if (G > 1) {
    // ---- class-group split: histogram of the survivors per class (every group of the image computes the same
    // one), then a deterministic longest-processing-time assignment of the classes to the G workgroups: classes
    // with more than 64 survivors, largest first, go to the least loaded group (load = survivors^2 / 64 ~ pair
    // tests); the rest are dealt out mod G.  The histogram also yields this group's survivor count.
    owner = reinterpret_cast<unsigned char*>(s_misc) + 2304;
    int* hist = reinterpret_cast<int*>(lkeys);                 // [C] (key area is free until the compaction)
    int* big = hist + p.C;                                     // [<= C] (count << 12 | class) in rank order
    for (int c = tid; c < p.C; c += blockDim.x) hist[c] = 0;
    if (tid == 0) s_misc[1] = 0;
    __syncthreads();
    for (int n0 = 0; n0 < p.N; n0 += blockDim.x) {             // wave-aggregated: one LDS atomic per (wave, class)
      const int n = n0 + tid;
      const bool act = n < p.N && scores[n] > p.conf_thr;
      const int cn = act ? clsb[n] : -1;
      u64 rem = __ballot(act);
      while (rem) {
        const int leader = __ffsll((long long)rem) - 1;
        const int c0 = __builtin_amdgcn_readlane(cn, leader);
        const u64 m = __ballot(cn == c0);
        if ((tid & 63) == leader) atomicAdd(&hist[c0], __popcll(m));
        rem &= ~m;
      }
    }
    __syncthreads();
    // rank of every big class in (count desc, class asc) order, computed in parallel: identical in all G workgroups
    for (int c = tid; c < p.C; c += blockDim.x) {
      owner[c] = (unsigned char)(c % G);
      const int hc = hist[c];
      if (hc > 64) {
        int rank = 0;
        for (int q = 0; q < p.C; ++q) {
          const int hq = hist[q];
          rank += (hq > 64 && (hq > hc || (hq == hc && q < c))) ? 1 : 0;
        }
        big[rank] = (hc << 12) | c;
        atomicAdd(&s_misc[1], 1);
      }
    }
    __syncthreads();
    if (tid == 0) {
      const int nb = s_misc[1];
      long load[4] = {0, 0, 0, 0};
      for (int i = 0; i < nb; ++i) {
        int q = 0;
        for (int t = 1; t < G; ++t) if (load[t] < load[q]) q = t;
        const long cnt = big[i] >> 12;
        load[q] += cnt * cnt;
        owner[big[i] & 4095] = (unsigned char)q;
      }
    }
    __syncthreads();
    int local = 0;
    for (int c = tid; c < p.C; c += blockDim.x) local += (owner[c] == g) ? hist[c] : 0;
    for (int d = 32; d > 0; d >>= 1) local += __shfl_xor(local, d);
    if ((tid & 63) == 0 && local) atomicAdd(&s_misc[0], local);
    __syncthreads();
    nsurv = s_misc[0];
    __syncthreads();                                            // hist / big (key area) are dead from here on
  } else {
    int local = 0;
    for (int n = tid; n < p.N; n += blockDim.x) local += (scores[n] > p.conf_thr) ? 1 : 0;
    for (int d = 32; d > 0; d >>= 1) local += __shfl_xor(local, d);
    if ((tid & 63) == 0 && local) atomicAdd(&s_misc[0], local);
    __syncthreads();
    nsurv = s_misc[0];
    if (nsurv == 0) {
      if (tid == 0) p.counts[b] = 0;
      return;
    }
  }
  int P = 64;
  while (P < nsurv) P <<= 1;
  // LDS: [lds_cap keys][4 ints]; the key slots beyond P are free -> survivors' boxes in sorted order
  float4* sbox = reinterpret_cast<float4*>(lkeys + P);
  if (P <= p.lds_cap) {
    if ((size_t)P * 8 + (size_t)nsurv * 16 <= (size_t)p.lds_cap * 8) yl_nms_run<true, true>(p, lkeys, P, nsurv, b, s_misc, sbox, owner);
    else yl_nms_run<true, false>(p, lkeys, P, nsurv, b, s_misc, nullptr, owner);
  } else {
    yl_nms_run<false, false>(p, p.gkeys + (size_t)b * p.gP, P, nsurv, b, s_misc, nullptr, owner);
  }
}

// Second half of the class-group split: one 256-thread workgroup per image turns the per-class kept lists the G
// NMS workgroups left behind into the ordered output (class ascending, score descending inside a class).
__global__ __launch_bounds__(256) void yl_nms_merge_kernel(YlNmsP p) {
  __shared__ int s_total;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int N = p.N, C = p.C, G = p.G;
  const float4* boxes = p.boxes + (size_t)b * N;
  const float* scores = p.scores + (size_t)b * N;
  int* ws_start = p.cls_ws + (size_t)b * 4 * C;
  int* ws_kept = ws_start + 2 * C;
  int* ws_off = ws_kept + C;
  const unsigned char* owner = reinterpret_cast<const unsigned char*>(p.done) + (size_t)b * 256;
  if (wave == 0) {
    int running = 0;
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int c = c0 + lane;
      const int v = (c < C) ? ws_kept[c] : 0;
      int incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      if (c < C) ws_off[c] = running + incl - v;
      running += __shfl(incl, 63);
    }
    if (lane == 0) s_total = running;
  }
  __syncthreads();
  float* dst = p.dets + (size_t)b * p.max_out * 6;
  int* dst_idx = p.keep_idx ? p.keep_idx + (size_t)b * p.max_out : nullptr;
  for (int c = wave; c < C; c += nwaves) {
    const int nk = ws_kept[c];
    if (nk <= 0) continue;
    const int off = ws_off[c];
    const int* src = p.kept_list + ((size_t)b * G + owner[c]) * N + ws_start[c];
    for (int k = lane; k < nk; k += 64) {
      const int orow = off + k;
      if (orow >= p.max_out) continue;
      const int idx = src[k];
      yl_write_det(p, b, dst, dst_idx, orow, boxes[idx], scores[idx], c, idx, true);
    }
  }
  if (tid == 0) p.counts[b] = s_total;
}

// ------------------------------------------------------------------------------------------------
// instance masks (build-defined, see include/yololite_hip.h): one block = 256 prototype pixels of one
// detection; coefficients staged in LDS; pixels outside the box never touch the prototypes.
__global__ __launch_bounds__(256) void yl_masks_kernel(YlLevels lv, const float* __restrict__ proto, int PH, int PW,
                                                       int NM, float scale_x, float scale_y,
                                                       const float4* __restrict__ boxes, const int* __restrict__ counts,
                                                       const int* __restrict__ keep_idx, int max_out, float thr,
                                                       unsigned char* __restrict__ masks) {
  __shared__ float coef[64];
  const int b = blockIdx.z, d = blockIdx.y;
  if (d >= counts[b] || d >= max_out) return;
  const int n = keep_idx[(size_t)b * max_out + d];
  if (threadIdx.x < NM) {
    const int l = yl_level_of(lv, n);
    const int nl = lv.A[l] * lv.S[l] * lv.S[l];
    coef[threadIdx.x] = lv.ptr[l][((size_t)b * nl + (n - lv.off[l])) * lv.E + 5 + lv.C + threadIdx.x];
  }
  __syncthreads();
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= PH * PW) return;
  const int y = pix / PW, x = pix - y * PW;
  const float4 bx = boxes[(size_t)b * lv.N + n];
  const float x1 = bx.x * scale_x, y1 = bx.y * scale_y, x2 = bx.z * scale_x, y2 = bx.w * scale_y;
  unsigned char out = 0;
  if ((float)x >= x1 && (float)x < x2 && (float)y >= y1 && (float)y < y2) {
    const float* pp = proto + (((size_t)b * PH + y) * PW + x) * NM;
    float acc = 0.0f;
    for (int k = 0; k < NM; ++k) acc = fmaf(coef[k], pp[k], acc);
    out = (yl_sigmoid(acc) > thr) ? 1 : 0;
  }
  masks[(((size_t)b * max_out + d) * PH + y) * PW + x] = out;
}

hipError_t yl_launch_masks(const YlLevels& lv, int B, const float* proto, int PH, int PW, int NM, int img_size,
                           const float4* boxes, const int* counts, const int* keep_idx, int max_out, float thr,
                           unsigned char* masks, hipStream_t st) {
  if (NM > 64) return hipErrorInvalidValue;
  dim3 grid((PH * PW + 255) / 256, max_out, B);
  hipLaunchKernelGGL(yl_masks_kernel, grid, dim3(256), 0, st, lv, proto, PH, PW, NM, (float)PW / (float)img_size,
                     (float)PH / (float)img_size, boxes, counts, keep_idx, max_out, thr, masks);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Instance masks at IMAGE resolution (BUILD-DEFINED, see include/yololite_hip.h: yl_masks_image): for every kept
// detection, sigmoid(mc . proto) on the prototype grid, bilinear (align_corners = false, edge-replicating like
// F.interpolate) sampling at the centre of every pixel of the ORIGINAL image mapped through the letterbox, crop to the
// detection's box, threshold -- one pass, no intermediate full-resolution probability map.
//
// The op is bound by the WRITE of the mask tensor (640 x 640 bit-packed: 51 KB per detection, ~330 detections per
// image), so the work is detection-major and every store is part of a contiguous run:
//   * work item = (image, detection, row part) -- `parts` = 1 unless the batch is small; ONE item list over the whole
//     batch (prefix of the per-image counts in LDS), workgroup g walks the items g, g + gridDim.x, ...: an image with
//     many detections is not the launch's tail;
//   * TWO launches of the same template walk the list.  BOXES = false: the rows of the item above and below the box
//     are ONE contiguous byte range each and are zero-filled straight from registers with 16-byte stores (16 VGPRs,
//     detection rows through the scalar cache: the waves never wait for their own stores; 545 MB in ~80 us at B = 32);
//   * BOXES = true: the rows that intersect the box are produced in tiles of <= 4 KB (whole rows; a row wider than the
//     tile is cut into column segments): the tile is zeroed in LDS at the SAME 16-byte phase as its global address, the
//     probabilities of the tile's prototype footprint (a few dozen prototype pixels: NM MACs + one sigmoid each, one
//     per thread) go to LDS, every half-wave takes one (row, 32-pixel word) unit -- a lane interpolates its pixel from
//     4 LDS values, `__ballot` packs the word -- and the finished tile leaves with 16-byte stores.
// Round 2's kernel was tile-major (a workgroup owned 64 x 4 pixels and looped over every detection): 4-byte words at
// a 51 KB stride, 3.8x write amplification, 3.1 ms per B = 32 launch; this pair takes 0.21 ms (box rows ~0.125 ms:
// a latency chain of ~6 us per detection at 5 workgroups per CU; zero fill ~0.08 ms).  The per-pixel arithmetic is
// unchanged (bitwise equal outputs, tools/masks_ab.py).
#define YL_MI_PCAP 2048          // probabilities of a tile's prototype footprint held in LDS
#define YL_MI_ROWS 128           // rows of a tile whose vertical taps are tabulated in LDS
struct YlMaskImgP {
  const float* proto; int PH, PW, NM;          // [B][PH][PW][NM]
  int S;                                        // network input size
  const float* dets; const int* counts; const int* keep_idx; int max_out;
  const float* backmap;                         // [B][5] padx,pady,scale,w0,h0 or nullptr (masks on the S x S input grid)
  const int* out_hw;                            // [B][2] output height, width
  const long long* mask_off;                    // [B] byte offset of image b's masks
  unsigned char* masks; float thr; int packed;
  int parts;                                    // row parts per detection (work items per detection)
  int tile;                                     // bytes of one output tile staged in LDS (multiple of 32)
  int B;
};

// loads through the constant address space: a wave-uniform address becomes a scalar (SMEM) load
#define YL_CONST_AS __attribute__((address_space(4)))
template <typename T> __device__ __forceinline__ const T YL_CONST_AS* yl_as_const(const T* q) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (const T YL_CONST_AS*)q;
#pragma clang diagnostic pop
}

// zero `len` bytes at g with the whole workgroup: bytes up to the first 16-byte boundary, 16-byte stores, tail bytes
__device__ __forceinline__ void yl_mi_zero(unsigned char* g, size_t len, int tid) {
  const size_t head = min((size_t)((16u - (unsigned)((uintptr_t)g & 15u)) & 15u), len);
  if ((size_t)tid < head) g[tid] = 0;
  const size_t body = (len - head) >> 4;
  uint4* gb = reinterpret_cast<uint4*>(g + head);
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = tid; i < body; i += 256) gb[i] = z;
  const size_t tail = len - head - (body << 4);
  if ((size_t)tid < tail) g[head + (body << 4) + tid] = 0;
}

#ifdef YL_MI_STAMP
// profiling aid (variant builds only: tools/build_variant.sh mistamp yl_post.hip -DYL_MI_STAMP): 100 MHz wall-clock ticks
// at the phase boundaries of the first items of ONE box-role and ONE fill-role workgroup
__device__ unsigned long long yl_mi_stamps[2 * 16 * 12];
__device__ unsigned long long yl_mi_blk[4096 * 2];        // start / end of every workgroup
extern "C" int yl_debug_mi_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(yl_mi_stamps), sizeof(yl_mi_stamps));
}
extern "C" int yl_debug_mi_blocks(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(yl_mi_blk), sizeof(yl_mi_blk));
}
#define YL_MIS(i) do { if (tid == 0 && role_i == 7 && nit < 16) yl_mi_stamps[((BOXES ? 0 : 1) * 16 + nit) * 12 + (i)] = wall_clock64(); } while (0)
#else
#define YL_MIS(i) do { } while (0)
#endif
// level tables of yl_masks_image_kernel in LDS: lptr[YL_MAX_LEVELS] (pointers), lnl[YL_MAX_LEVELS], loff[YL_MAX_LEVELS + 1];
// ONE definition for the kernel's carve-up and the launcher's dynamic-LDS request (a hand-counted
// "YL_MAX_LEVELS * 12 + 8" was 28 bytes short and pre[B-6..B] lay past the request)
#define YL_MI_LEVEL_TABLE_BYTES (YL_MAX_LEVELS * (sizeof(const float*) + sizeof(int)) + (YL_MAX_LEVELS + 1) * sizeof(int))
template <bool BOXES>
__global__ __launch_bounds__(256, BOXES ? 5 : 8) void yl_masks_image_kernel(YlLevels lv, YlMaskImgP p) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) unsigned char mi_smem[];
  float* coef = reinterpret_cast<float*>(mi_smem);                               // [64]
  float* pbuf = coef + 64;                                                       // [YL_MI_PCAP]
  float* rtab = pbuf + YL_MI_PCAP;                                               // [YL_MI_ROWS][4] vertical taps of the tile's rows
  unsigned char* tile = reinterpret_cast<unsigned char*>(rtab + 4 * YL_MI_ROWS); // [tile + 32]
  // level table in LDS: indexing the by-value YlLevels dynamically keeps the whole struct in SGPRs (106 SGPRs, 88 of
  // them spilled into VGPR lanes -> 106 VGPRs -> 4 workgroups per CU instead of 8)
  const float** lptr = reinterpret_cast<const float**>(tile + p.tile + 32);      // [YL_MAX_LEVELS]
  int* lnl = reinterpret_cast<int*>(lptr + YL_MAX_LEVELS);                       // [YL_MAX_LEVELS] A * S * S
  int* loff = lnl + YL_MAX_LEVELS;                                               // [YL_MAX_LEVELS + 1]
  int* pre = reinterpret_cast<int*>(tile + p.tile + 32 + YL_MI_LEVEL_TABLE_BYTES); // [B + 1] first item of every image
  if (BOXES && threadIdx.x == 0) {
#pragma unroll
    for (int l = 0; l < YL_MAX_LEVELS; ++l) {
      lptr[l] = lv.ptr[l]; lnl[l] = lv.A[l] * lv.S[l] * lv.S[l];
      loff[l] = l <= lv.L ? lv.off[l] : 0x7fffffff;
    }
    loff[YL_MAX_LEVELS] = lv.L == YL_MAX_LEVELS ? lv.off[YL_MAX_LEVELS] : 0x7fffffff;
    for (int l = lv.L + 1; l <= YL_MAX_LEVELS; ++l) loff[l] = 0x7fffffff;
  }
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef YL_MI_STAMP
  if (tid == 0 && blockIdx.x < 2048) yl_mi_blk[2 * (blockIdx.x + (BOXES ? 2048 : 0))] = wall_clock64();
#endif
  const int NM = p.NM, PW = p.PW, PH = p.PH;
  const float kx = (float)PW / (float)p.S, ky = (float)PH / (float)p.S;
  // items of every image -> exclusive prefix in LDS (one wave scans it; B is small).  Walking the images one scalar
  // load chain at a time cost every workgroup ~1.5 us per image (B = 32: 50 us of a 240 us launch)
  for (int i = tid; i < p.B; i += 256) {
    const int h = p.out_hw[2 * i], w = p.out_hw[2 * i + 1];
    pre[i + 1] = (h > 0 && w > 0) ? max(min(p.counts[i], p.max_out), 0) * p.parts : 0;
  }
  if (tid == 0) pre[0] = 0;
  __syncthreads();
  if (tid < 64) {
    int carry = 0;
    for (int c0 = 0; c0 < p.B; c0 += 64) {
      int v = (c0 + lane < p.B) ? pre[c0 + lane + 1] : 0;
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v += t;
      }
      v += carry;
      if (c0 + lane < p.B) pre[c0 + lane + 1] = v;
      carry = __shfl(v, 63);
    }
  }
  __syncthreads();
  const long long total = pre[p.B];
  // per-image state (workgroup-uniform), refreshed when the item walk enters another image
  int b = -1, base = 0;                         // image, first global item of image b
  int H = 0, W = 0, wrow = 0, nsegs = 1, RB = 1;
  float sx = 1.0f, padx = 0.0f, pady = 0.0f;
  unsigned char* mb = nullptr;
  const float* pimg = nullptr;
  // prototype-grid coordinate of a pixel centre: letterbox coordinate (c + 0.5) * scale + pad, then the
  // align_corners = false source index of a PW/S resize, clamped at 0 (F.interpolate); i0 / i1 = the two taps
  auto taps = [&](int c, float pad, float k, int lim, int& i0, int& i1, float& w) {
    const float f = fmaxf(fmaxf(((float)c + 0.5f) * sx + pad, 0.0f) * k - 0.5f, 0.0f);
    i0 = min((int)floorf(f), lim - 1);
    i1 = min(i0 + 1, lim - 1);
    w = f - (float)i0;
  };
  // ONE item list over the whole batch (image-major, then detection, then row part), dealt round-robin to the
  // workgroups.  Why two instantiations instead of one kernel doing both (the first version): the uniform state of
  // the two paths together needed 106 SGPRs + 88 spilled into VGPR lanes (4 workgroups per CU), and gfx9 counts loads
  // and stores in ONE in-order counter (vmcnt), so a workgroup's box path (detection row -> candidate index ->
  // coefficients, prototype footprint -> LDS -> stores) waited for the acknowledgement of its own 51 KB of zero
  // stores: 0.24-0.26 ms per B = 32 launch.  The two launches write disjoint bytes.
  constexpr bool boxes = BOXES, fill = !BOXES;
  const long long role_i = blockIdx.x, role_n = gridDim.x;
  {
  int nit = -1;
  for (long long it = role_i; it < total; it += role_n) {
    ++nit;
    YL_MIS(0);
    if (b < 0 || it >= pre[b + 1]) {            // another image: its geometry through the scalar cache
      do { ++b; } while (it >= pre[b + 1]);
      base = pre[b];
      H = yl_as_const(p.out_hw)[2 * b]; W = yl_as_const(p.out_hw)[2 * b + 1];
      wrow = p.packed ? ((W + 31) >> 5) << 2 : W;                                // bytes per mask row
      nsegs = max((wrow + p.tile - 1) / p.tile, 1);                              // column segments of a row (1 unless huge)
      RB = nsegs > 1 ? 1 : min(max(1, p.tile / max(wrow, 1)), YL_MI_ROWS);       // rows per tile
      sx = 1.0f; padx = 0.0f; pady = 0.0f;
      if (p.backmap) {
        const float YL_CONST_AS* bm = yl_as_const(p.backmap + 5 * b);
        padx = bm[0]; pady = bm[1]; sx = bm[2];
      }
      mb = p.masks + yl_as_const(p.mask_off)[b];
      pimg = p.proto + (size_t)b * PH * PW * NM;
    }
    const int li = (int)(it - base);
    const int d = li / p.parts, part = li - d * p.parts;
    const int R0 = (int)((long long)H * part / p.parts), R1 = (int)((long long)H * (part + 1) / p.parts);   // rows [R0, R1)
    if (R0 >= R1) continue;
    // the detection row through the SCALAR cache (uniform address): lgkmcnt, not vmcnt, so the fill role never waits
    // for its own stores
    const float YL_CONST_AS* dr = yl_as_const(p.dets + ((size_t)b * p.max_out + d) * 6);
    const float x1 = dr[0], y1 = dr[1], x2 = dr[2], y2 = dr[3];
    unsigned char* md = mb + (size_t)d * H * wrow;
    // integer hull of the box (a superset: the per-pixel predicate below decides, so NaN / huge values are harmless)
    const int xlo = (int)fminf(fmaxf(floorf(x1), 0.0f), (float)W), xhi = (int)fminf(fmaxf(floorf(x2), -1.0f), (float)(W - 1));
    const int ylo = (int)fminf(fmaxf(floorf(y1), 0.0f), (float)H), yhi = (int)fminf(fmaxf(floorf(y2), -1.0f), (float)(H - 1));
    const int ya = max(R0, ylo), yb = min(R1 - 1, yhi);
    const bool hit = ya <= yb && xlo <= xhi;
    YL_MIS(1);
    if (fill) {
      if (!hit) {                                                                // no box pixel in these rows
        yl_mi_zero(md + (size_t)R0 * wrow, (size_t)(R1 - R0) * wrow, tid);
      } else {
        if (ya > R0) yl_mi_zero(md + (size_t)R0 * wrow, (size_t)(ya - R0) * wrow, tid);
        if (yb + 1 < R1) yl_mi_zero(md + (size_t)(yb + 1) * wrow, (size_t)(R1 - 1 - yb) * wrow, tid);
      }
    }
    YL_MIS(2);
    if (!boxes || !hit) continue;
    if (tid < NM) {           // safe: every earlier read of coef[] is followed by a workgroup barrier
      const int cand = yl_as_const(p.keep_idx)[(size_t)b * p.max_out + d];
      int l = 0;
      while (cand >= loff[l + 1]) ++l;
      coef[tid] = lptr[l][((size_t)b * lnl[l] + (cand - loff[l])) * lv.E + 5 + lv.C + tid];
    }
    for (int seg = 0; seg < nsegs; ++seg) {
      const int c0 = seg * p.tile, c1 = min(c0 + p.tile, wrow) - 1;      // bytes [c0, c1] of every row
      const int px0 = p.packed ? c0 * 8 : c0, px1 = min(p.packed ? c1 * 8 + 7 : c1, W - 1);
      const int xa = max(px0, xlo), xb = min(px1, xhi);
      if (xa > xb) {                                                             // only with nsegs > 1
        for (int r = ya; r <= yb; ++r) yl_mi_zero(md + (size_t)r * wrow + c0, (size_t)(c1 - c0 + 1), tid);
        continue;
      }
      int pu0, pu1, tmp; float tw;
      taps(xa, padx, kx, PW, pu0, tmp, tw);
      taps(xb, padx, kx, PW, tmp, pu1, tw);
      const int pw = pu1 - pu0 + 1;
      const int w0 = xa >> 5, nw = (xb >> 5) - w0 + 1;                           // 32-pixel words that hold box pixels
      for (int t0 = ya; t0 <= yb;) {
        int t1 = min(yb, t0 + RB - 1);
        int pv0, pv1;
        taps(t0, pady, ky, PH, pv0, tmp, tw);
        for (;;) {                                  // shrink the row group until its footprint fits (one row always
          taps(t1, pady, ky, PH, tmp, pv1, tw);     // does: 2 * PW <= YL_MI_PCAP is checked by the launcher)
          if ((pv1 - pv0 + 1) * pw <= YL_MI_PCAP || t1 == t0) break;
          t1 = t0 + ((t1 - t0) >> 1);
        }
        const int np = (pv1 - pv0 + 1) * pw;
        unsigned char* g0 = md + (size_t)t0 * wrow + c0;
        const int len = nsegs > 1 ? (c1 - c0 + 1) : (t1 - t0 + 1) * wrow;
        const int mis = (int)((uintptr_t)g0 & 15u);
        YL_MIS(3);
        __syncthreads();                                  // A: the previous tile has left LDS; coef[] is written
        YL_MIS(4);
        for (int i = tid; i < ((mis + len + 15) >> 4); i += 256) reinterpret_cast<uint4*>(tile)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < t1 - t0 + 1) {                          // vertical taps of the tile's rows, as pbuf row offsets
          int v0, v1; float lvw;
          taps(t0 + tid, pady, ky, PH, v0, v1, lvw);
          reinterpret_cast<float4*>(rtab)[tid] = make_float4(__int_as_float((v0 - pv0) * pw - pu0), __int_as_float((v1 - pv0) * pw - pu0),
                                                             lvw, 0.0f);
        }
        {
          const float rpw = 1.0f / (float)pw;
          for (int t = tid; t < np && tid < 192; t += 192) {          // waves 0-2 (wave 3 is the store wave, see below)
            int q = (int)((float)t * rpw);
            int r = t - q * pw;
            if (r < 0) { --q; r += pw; } else if (r >= pw) { ++q; r -= pw; }
            const float4* pp = reinterpret_cast<const float4*>(pimg + ((size_t)(pv0 + q) * PW + (pu0 + r)) * NM);
            float acc = 0.0f;
            if (NM == 32) {                               // the build's default: all eight loads in flight at once
              float4 v[8];
#pragma unroll
              for (int k = 0; k < 8; ++k) v[k] = pp[k];
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                acc = fmaf(coef[4 * k], v[k].x, acc); acc = fmaf(coef[4 * k + 1], v[k].y, acc);
                acc = fmaf(coef[4 * k + 2], v[k].z, acc); acc = fmaf(coef[4 * k + 3], v[k].w, acc);
              }
            } else if ((NM & 3) == 0) {
              for (int k = 0; k < NM; k += 4) {
                const float4 v = pp[k >> 2];
                acc = fmaf(coef[k], v.x, acc); acc = fmaf(coef[k + 1], v.y, acc);
                acc = fmaf(coef[k + 2], v.z, acc); acc = fmaf(coef[k + 3], v.w, acc);
              }
            } else {
              const float* ps = reinterpret_cast<const float*>(pp);
              for (int k = 0; k < NM; ++k) acc = fmaf(coef[k], ps[k], acc);
            }
            pbuf[t] = yl_sigmoid(acc);
          }
        }
        YL_MIS(5);
        __syncthreads();                                  // B: probabilities + zeroed tile visible
        YL_MIS(6);
        const int nunits = (t1 - t0 + 1) * nw;
        // half-wave h of wave w takes the units 2w + h, 2w + h + 8, ...: (row q, word r) advanced incrementally
        int u = (tid >> 6) * 2 + (lane >> 5);
        int q = u / nw, r = u - q * nw;
        const int dq = 8 / nw, dr_ = 8 - dq * nw;
        for (int ub = (tid >> 6) * 2; ub < nunits; ub += 8) {                    // wave-uniform: __ballot below
          const bool valid = u < nunits;
          const int y = t0 + q, xw = w0 + r, x = (xw << 5) + (lane & 31);
          bool on = false;
          if (valid && x <= px1 && (float)x >= x1 && (float)x < x2 && (float)y >= y1 && (float)y < y2) {
            int u0, u1; float lu;
            taps(x, padx, kx, PW, u0, u1, lu);
            const float4 rt = reinterpret_cast<const float4*>(rtab)[q];
            const float lv_ = rt.z;
            const float* r0p = pbuf + __float_as_int(rt.x);
            const float* r1p = pbuf + __float_as_int(rt.y);
            const float p00 = r0p[u0], p01 = r0p[u1], p10 = r1p[u0], p11 = r1p[u1];
            const float top = p00 + (p01 - p00) * lu, bot = p10 + (p11 - p10) * lu;
            on = (top + (bot - top) * lv_) > p.thr;
          }
          unsigned char* trow = tile + mis + (size_t)q * (nsegs > 1 ? 0 : wrow) - c0;
          if (p.packed) {
            const unsigned long long bits = __ballot(on);
            if (valid && (lane & 31) == 0) *reinterpret_cast<unsigned*>(trow + (xw << 2)) = (unsigned)(bits >> (lane & 32));
          } else if (valid && x <= px1) {
            trow[x] = on ? 1 : 0;
          }
          u += 8; q += dq; r += dr_;
          if (r >= nw) { r -= nw; ++q; }
        }
        YL_MIS(7);
        __syncthreads();                                  // C: tile complete
        YL_MIS(8);
        // Only wave 3 stores (and it never issues a vector load): vmcnt counts loads and stores in ONE in-order
        // counter, so a wave that waits for a load also waits for the acknowledgement of its earlier stores.
        if (tid >= 192) {
          const int l = tid - 192;
          const int head = min((16 - mis) & 15, len);
          if (l < head) g0[l] = tile[mis + l];
          const int body = (len - head) >> 4;
          const uint4* src = reinterpret_cast<const uint4*>(tile + mis + head);
          uint4* gb = reinterpret_cast<uint4*>(g0 + head);
          for (int i = l; i < body; i += 64) gb[i] = src[i];
          const int tail = len - head - (body << 4);
          if (l < tail) g0[head + (body << 4) + l] = tile[mis + head + (body << 4) + l];
        }
        YL_MIS(9);
        t0 = t1 + 1;
      }
    }
  }
  }
#ifdef YL_MI_STAMP
  if (tid == 0 && blockIdx.x < 2048) yl_mi_blk[2 * (blockIdx.x + (BOXES ? 2048 : 0)) + 1] = wall_clock64();
#endif
}

hipError_t yl_launch_masks_image(const YlLevels& lv, int B, const float* proto, int PH, int PW, int NM, int S,
                                 const float* dets, const int* counts, const int* keep_idx, int max_out, float thr,
                                 const float* backmap, const int* out_hw, const long long* mask_off, int max_h, int max_w,
                                 int packed, unsigned char* masks, hipStream_t st) {
  if (NM > 64 || max_h < 1 || max_w < 1 || 2 * PW > YL_MI_PCAP || B > 8192) return hipErrorInvalidValue;
  YlMaskImgP p;
  p.proto = proto; p.PH = PH; p.PW = PW; p.NM = NM; p.S = S; p.dets = dets; p.counts = counts; p.keep_idx = keep_idx;
  p.max_out = max_out; p.backmap = backmap; p.out_hw = out_hw; p.mask_off = mask_off; p.masks = masks; p.thr = thr;
  p.packed = packed;
  // small batches: cut every detection's rows into parts so that a handful of detections still spreads over the chip
  p.parts = B >= 16 ? 1 : min(max(32 / B, 1), min(16, max_h));
  p.B = B;
  const int blocks_fill = 2048, blocks_box = 2048;      // 8 x 256 CUs each (tuned: tools/masks_ab.py)
  const long long most = (long long)B * max_out * p.parts;
  const unsigned gfill = (unsigned)min((long long)blocks_fill, most), gbox = (unsigned)min((long long)blocks_box, most);
  // bit-packed rows are short (80 B at 640 px): 4 KB tiles hold ~50 rows and keep 8 workgroups per CU; uint8 rows get 16 KB
  p.tile = packed ? 4096 : 16384;
  // the kernel's carve-up (mi_smem): coef[64] pbuf[PCAP] rtab[ROWS][4] | tile[tile + 32] | lptr[L] lnl[L] loff[L + 1] | pre[B + 1]
  const size_t lds = (size_t)(64 + YL_MI_PCAP + 4 * YL_MI_ROWS) * sizeof(float) + p.tile + 32 + YL_MI_LEVEL_TABLE_BYTES +
                     (size_t)(B + 1) * sizeof(int);
  hipLaunchKernelGGL(yl_masks_image_kernel<true>, dim3(gbox), dim3(256), lds, st, lv, p);
  hipLaunchKernelGGL(yl_masks_image_kernel<false>, dim3(gfill), dim3(256), lds, st, lv, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
static int g_nms_lds_max = 0;

hipError_t yl_post_init() {
  // 128 KiB of keys + scratch: needs the opt-in above the default 64 KiB dynamic-LDS limit
  const int want = YL_LDS_KEYS_MAX * 8 + YL_NMS_SCRATCH;
  hipError_t e = hipFuncSetAttribute((const void*)yl_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want);
  if (e != hipSuccess) return e;
  g_nms_lds_max = want;
  return hipSuccess;
}

hipError_t yl_launch_decode_score(const YlLevels& lv, int B, const YlDecodeP& p, hipStream_t st) {
  dim3 grid((lv.N + 127) / 128, B), block(128);
  const size_t lds = (size_t)2 * 64 * (lv.E | 1) * sizeof(float);
  if (lds <= 60 * 1024)
    hipLaunchKernelGGL(yl_decode_score_kernel<true>, grid, block, lds, st, lv, B, p);
  else
    hipLaunchKernelGGL(yl_decode_score_kernel<false>, grid, block, 0, st, lv, B, p);
  return hipGetLastError();
}

hipError_t yl_launch_decode_only(const YlLevels& lv, int B, int center_mode, int wh_mode, float* box, float* obj,
                                 float* cls, hipStream_t st) {
  dim3 g1((lv.N + 255) / 256, B);
  hipLaunchKernelGGL(yl_decode_box_kernel, g1, dim3(256), 0, st, lv, B, center_mode, wh_mode, (float4*)box, obj);
  if (lv.C > 0) {
    dim3 g2((unsigned)(((size_t)lv.N * lv.C + 255) / 256), B);
    hipLaunchKernelGGL(yl_copy_cls_kernel, g2, dim3(256), 0, st, lv, B, cls);
  }
  return hipGetLastError();
}

hipError_t yl_launch_nms(const YlNmsP& p, int B, hipStream_t st) {
  const size_t lds = (size_t)p.lds_cap * 8 + YL_NMS_SCRATCH;
  if ((int)lds > g_nms_lds_max && lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(yl_nms_kernel, dim3(B, p.G > 1 ? p.G : 1), dim3(1024), lds, st, p);
  if (p.G > 1) hipLaunchKernelGGL(yl_nms_merge_kernel, dim3(B), dim3(256), 0, st, p);
  return hipGetLastError();
}
