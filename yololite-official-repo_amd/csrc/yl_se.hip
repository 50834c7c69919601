// Squeeze-excite gate (YL_OP_SE) for gfx950: timm's SqueezeExcite inside the `ir_..._se0.25` blocks of the
// tf_efficientnetv2_b* backbones the reference builds through timm.create_model (model_v2.py:94-100;
// configs/v2_models/yololite_{n,s,m}.yaml):
//     x_se = x.mean((2, 3), keepdim=True); x_se = act(conv_reduce(x_se)); gate = sigmoid(conv_expand(x_se)); x * gate
// The multiply itself happens in the B-operand path of the 1x1 projection conv that follows (YlConvP::scale) -- the
// expanded tensor is read ONCE more for the mean and never rewritten.
//
// Deterministic: the spatial mean is a two-pass sum in a fixed order (P partial sums per image and channel over
// contiguous pixel ranges, then the P partials in index order) -- no floating-point atomics, bitwise repeatable and
// independent of the batch an image is part of.
//
//   yl_se_pool_kernel   grid (P, B) x 256 threads: workgroup (p, b) sums pixels [p*HW/P, (p+1)*HW/P) of image b for all C
//                       channels; thread = (pixel lane, float4 channel group), coalesced 16-byte loads along C, the
//                       pixel lanes reduced through LDS in lane order.  HBM-bound: one read of the tensor.
//                       Only a fallback: when the tensor comes from a stand-alone depthwise launch (always, in the
//                       efficientnetv2 blocks) that launch leaves the partial sums itself (yl_dw_tile_kernel<.., POOL>).
//   yl_se_gate_kernel   grid (B) x 1024 threads: mean = (sum of the P partials) / HW; reduce FC (one wave per output,
//                       lanes across C, butterfly sum) + activation; expand FC (thread per channel, RD <= 256 terms) +
//                       sigmoid.  Latency-bound, a few KB per image.
#include "yl_internal.h"
#include "yl_dev.h"
#include "yl_decode.h"

#define YL_SE_MAXC 4096          // channels (LDS: 16 KiB of means)
#define YL_SE_MAXRD 256

// partial sums per image: enough workgroups to pull the tensor at HBM speed even for one image (a workgroup streams
// ~64 pixels x C at a time), at most 64 (the second pass walks them serially), at least 1
int yl_se_parts(int HW, int C) {
  (void)C;
  int p = (HW + 63) / 64;
  if (p > 64) p = 64;
  if (p < 1) p = 1;
  return p;
}

__global__ __launch_bounds__(256) void yl_se_pool_kernel(YlSeP p) {
  __shared__ __attribute__((aligned(16))) float red[256 * 4];
  const int tid = threadIdx.x;
  const int part = blockIdx.x, b = blockIdx.y;
  const int C4 = p.C >> 2;
  const int G = C4 < 256 ? C4 : 256;             // channel groups handled side by side
  const int PL = 256 / G;                         // pixel lanes
  const int pix0 = (int)((long)p.HW * part / p.P), pix1 = (int)((long)p.HW * (part + 1) / p.P);
  const int pl = tid / G, cg0 = tid - pl * G;
  const float* xb = p.x + (size_t)b * p.HW * p.C;
  float* out = p.partial + ((size_t)b * p.P + part) * p.C;
  for (int cgb = 0; cgb < C4; cgb += G) {         // C > 1024: several passes over the pixel range
    const int cg = cgb + cg0;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (pl < PL && cg < C4)
      for (int i = pix0 + pl; i < pix1; i += PL) s += yl_ld4(xb + (size_t)i * p.C + 4 * cg);
    if (PL == 1) {                                 // (threads beyond G have pl == 1: they hold zeros and must not store)
      if (pl == 0 && cg < C4) *reinterpret_cast<f32x4*>(out + 4 * cg) = s;
      continue;
    }
    __syncthreads();
    if (pl < PL) *reinterpret_cast<f32x4*>(red + 4 * tid) = s;
    __syncthreads();
    if (pl == 0 && cg < C4) {
      f32x4 t = s;
      for (int q = 1; q < PL; ++q) t += *reinterpret_cast<const f32x4*>(red + 4 * (q * G + cg0));   // lane order: fixed
      *reinterpret_cast<f32x4*>(out + 4 * cg) = t;
    }
  }
}

// 1024 threads per image: the mean (P partials in index order), the reduce FC (one wave per output row at a time, lanes
// across C, butterfly sum: a fixed order) and the expand FC (thread per channel; w2 is stored TRANSPOSED [RD][C] so the RD
// reads of a wave are coalesced rows).
__global__ __launch_bounds__(1024) void yl_se_gate_kernel(YlSeP p) {
  __shared__ float mean[YL_SE_MAXC];
  __shared__ float rd[YL_SE_MAXRD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const float* part = p.partial + (size_t)b * p.P * p.C;
  const float hw = (float)p.HW;
  // (round 6) The three loops below are chains of DEPENDENT adds over independent loads with run-time trip counts: compiled
  // rolled, every iteration waited for its own L2 round trip -- 50 us per launch for a few KB of data, 20 launches per
  // efficientnetv2 forward.  Each loop now fetches a batch of 8 operands first and then adds them IN THE SAME ORDER: the
  // same fp32 bits, the latencies overlapped.
  for (int c = tid; c < p.C; c += 1024) {
    float s = part[c];
    int q = 1;
    for (; q + 8 <= p.P; q += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(q + u) * p.C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; q < p.P; ++q) s += part[(size_t)q * p.C + c];
    mean[c] = s / hw;
  }
  __syncthreads();
  for (int j = wave; j < p.RD; j += 16) {
    const float* w = p.w1 + (size_t)j * p.C;
    float s = 0.0f;
    int c = lane;
    for (; c + 7 * 64 < p.C; c += 8 * 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = w[c + 64 * u];
#pragma unroll
      for (int u = 0; u < 8; ++u) s = fmaf(v[u], mean[c + 64 * u], s);
    }
    for (; c < p.C; c += 64) s = fmaf(w[c], mean[c], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) rd[j] = yl_act1(s + p.b1[j], p.act);
  }
  __syncthreads();
  for (int c = tid; c < p.C; c += 1024) {
    float s = 0.0f;
    int j = 0;
    for (; j + 8 <= p.RD; j += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p.w2[(size_t)(j + u) * p.C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s = fmaf(v[u], rd[j + u], s);
    }
    for (; j < p.RD; ++j) s = fmaf(p.w2[(size_t)j * p.C + c], rd[j], s);
    p.gate[(size_t)b * p.C + c] = yl_sigmoid(s + p.b2[c]);
  }
}

hipError_t yl_launch_se(const YlSeP& p, bool pooled, hipStream_t st) {
  if (p.C < 4 || (p.C & 3) || p.C > YL_SE_MAXC || p.RD < 1 || p.RD > YL_SE_MAXRD || p.HW < 1 || p.P < 1 || p.B < 1)
    return hipErrorInvalidValue;
  if (!pooled) hipLaunchKernelGGL(yl_se_pool_kernel, dim3(p.P, p.B), dim3(256), 0, st, p);
  hipLaunchKernelGGL(yl_se_gate_kernel, dim3(p.B), dim3(1024), 0, st, p);
  return hipGetLastError();
}
