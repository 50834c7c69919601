// Element-wise / reduction ops of the hgnetv2 and convnextv2 feature extractors (ABI v5: YL_OP_POOL, YL_OP_COPY,
// YL_OP_LN, YL_OP_GRN, YL_OP_NHWC4) -- timm models/hgnet.py (StemV2, HighPerfGpuBlock) and models/convnext.py
// (LayerNorm2d, ConvNeXtBlock, GlobalResponseNormMlp) behind /root/reference/scripts/model/model_v2.py:94-100,266-272
// (configs/models/edge_xl.yaml:4, configs/v2_models/yololite_l.yaml:4).  All of them move every byte once: NHWC fp32,
// consecutive lanes = consecutive channel quads of one pixel (16-byte coalesced accesses), HBM-bound.
#include "yl_internal.h"
#include "yl_dev.h"

// ---- activation pass: GELU (erf form, torch's operation order x * 0.5 * (1 + erf(x / sqrt 2))) or ReLU + timm's
// LearnableAffineBlock (scale * relu(v) + bias, two roundings), then the residual -- in place over a conv's output -------
__device__ __forceinline__ float yl_post1(float v, int act, float lab_s, float lab_b) {
  if (act == YL_ACT_GELU) return (v * 0.5f) * (1.0f + erff(v * 0.70710678118654752440f));
  v = fmaxf(v, 0.0f);
  return lab_s * v + lab_b;
}
__global__ __launch_bounds__(256) void yl_act_kernel(YlOpP p) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const size_t total = (size_t)p.B * p.OH * p.OW * p.C;
  if (i >= total) return;
  f32x4 v = yl_ld4(p.x + i);
  v.x = yl_post1(v.x, p.act, p.lab_s, p.lab_b); v.y = yl_post1(v.y, p.act, p.lab_s, p.lab_b);
  v.z = yl_post1(v.z, p.act, p.lab_s, p.lab_b); v.w = yl_post1(v.w, p.act, p.lab_s, p.lab_b);
  if (p.res) v += yl_ld4(p.res + i);
  *reinterpret_cast<f32x4*>(p.out + i) = v;
}

// ---- max-pool over the zero-extended input (StemV2: F.pad(x, (0,1,0,1)) -> MaxPool2d(kernel 2, stride 1)) -------------
__global__ __launch_bounds__(256) void yl_pool_kernel(YlOpP p) {
  const int cq = p.C >> 2;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)p.B * p.OH * p.OW * cq) return;
  const size_t lin = i / cq;
  const int c = (int)(i - lin * cq) * 4;
  const int ohw = p.OH * p.OW;
  const int b = (int)(lin / ohw);
  const int rem = (int)(lin - (size_t)b * ohw);
  const int oy = rem / p.OW, ox = rem - oy * p.OW;
  const int y0 = oy * p.stride - p.pad_t, x0 = ox * p.stride - p.pad_l;
  const float* xb = p.x + (size_t)b * p.H * p.W * p.C + c;
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (int dy = 0; dy < p.k; ++dy)
    for (int dx = 0; dx < p.k; ++dx) {
      const int iy = y0 + dy, ix = x0 + dx;
      const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const f32x4 v = in ? yl_ld4(xb + ((size_t)iy * p.W + ix) * p.C) : (f32x4){0.f, 0.f, 0.f, 0.f};
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  *reinterpret_cast<f32x4*>(p.out + lin * p.C + c) = m;
}

// ---- channel-slice copy (one input of a torch.cat along the channels) ------------------------------------------------
__global__ __launch_bounds__(256) void yl_copy_kernel(YlOpP p) {
  const int cq = p.C >> 2;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)p.B * p.OH * p.OW * cq) return;
  const size_t lin = i / cq;
  const int c = (int)(i - lin * cq) * 4;
  *reinterpret_cast<f32x4*>(p.out + lin * p.ldo + p.ch_off + c) = yl_ld4(p.x + lin * p.C + c);
}

// ---- LayerNorm over C per pixel: one wave per pixel, two-pass moments (mean, then the biased variance of the centred
// values -- the textbook form; torch's RowwiseMoments differs in summation order only), lanes over channel quads ---------
__global__ __launch_bounds__(256) void yl_ln_kernel(YlOpP p) {
  const int lane = threadIdx.x & 63;
  const size_t pix = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (size_t)p.B * p.OH * p.OW) return;
  const int cq = p.C >> 2;
  const float* xr = p.x + pix * p.C;
  float s = 0.f;
  for (int q = lane; q < cq; q += 64) { const f32x4 v = yl_ld4(xr + 4 * q); s += (v.x + v.y) + (v.z + v.w); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)p.C;
  float q2 = 0.f;
  for (int q = lane; q < cq; q += 64) {
    const f32x4 v = yl_ld4(xr + 4 * q);
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    q2 += (a * a + b * b) + (c * c + d * d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o, 64);
  const float rstd = 1.0f / sqrtf(q2 / (float)p.C + p.eps);
  float* orow = p.out + pix * p.C;
  for (int q = lane; q < cq; q += 64) {
    const f32x4 v = yl_ld4(xr + 4 * q), w = yl_ld4(p.w + 4 * q), bb = yl_ld4(p.b + 4 * q);
    f32x4 r;
    r.x = (v.x - mean) * rstd * w.x + bb.x; r.y = (v.y - mean) * rstd * w.y + bb.y;
    r.z = (v.z - mean) * rstd * w.z + bb.z; r.w = (v.w - mean) * rstd * w.w + bb.w;
    *reinterpret_cast<f32x4*>(orow + 4 * q) = r;
  }
}

// ---- GlobalResponseNorm gate.  Pass 1: partial[b][r][c] = sum of x^2 over every P-th pixel of image b (thread = channel
// quad, pixels strided by P: a fixed function of the shape -> bitwise repeatable, no float atomics).  Pass 2 (one
// workgroup per image): g[c] = sqrt(sum of the P partials in index order), mean over C by a fixed-shape tree, gate.
__global__ __launch_bounds__(256) void yl_grn_sumsq_kernel(YlOpP p) {
  const int cq = p.C >> 2;
  const int b = blockIdx.z, r = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= cq) return;
  const int HW = p.OH * p.OW;
  const float* xb = p.x + (size_t)b * HW * p.C + 4 * q;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int i = r; i < HW; i += p.P) {
    const f32x4 v = yl_ld4(xb + (size_t)i * p.C);
    s.x = fmaf(v.x, v.x, s.x); s.y = fmaf(v.y, v.y, s.y); s.z = fmaf(v.z, v.z, s.z); s.w = fmaf(v.w, v.w, s.w);
  }
  *reinterpret_cast<f32x4*>(p.partial + ((size_t)b * p.P + r) * p.C + 4 * q) = s;
}

__global__ __launch_bounds__(1024) void yl_grn_gate_kernel(YlOpP p) {
  __shared__ float red[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* pb = p.partial + (size_t)b * p.P * p.C;
  float loc = 0.f;
  for (int c = tid; c < p.C; c += 1024) {
    float s = 0.f;
    for (int r = 0; r < p.P; ++r) s += pb[(size_t)r * p.C + c];
    const float g = sqrtf(s);
    p.out[(size_t)b * p.C + c] = g;                     // parked in the gate row until the mean is known
    loc += g;
  }
  red[tid] = loc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float den = red[0] / (float)p.C + p.eps;
  for (int c = tid; c < p.C; c += 1024) {
    const float g = p.out[(size_t)b * p.C + c];
    p.out[(size_t)b * p.C + c] = 1.0f + p.w[c] * (g / den);
  }
}

// ---- network input NCHW [B,3,S,S] -> NHWC [B,S,S,4] (channel 3 = 0) --------------------------------------------------
__global__ __launch_bounds__(256) void yl_nhwc4_kernel(YlOpP p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t plane = (size_t)p.H * p.W;
  if (i >= (size_t)p.B * plane) return;
  const size_t b = i / plane, r = i - b * plane;
  const float* xb = p.x + b * 3 * plane + r;
  const f32x4 v = {xb[0], xb[plane], xb[2 * plane], 0.0f};
  *reinterpret_cast<f32x4*>(p.out + 4 * i) = v;
}

int yl_grn_parts(int HW) {
  int P = (HW + 63) / 64;                 // >= 64 pixels per partial
  if (P > 64) P = 64;
  if (P < 1) P = 1;
  return P;
}

hipError_t yl_launch_op(int op, const YlOpP& p, hipStream_t st) {
  const size_t pix = (size_t)p.B * p.OH * p.OW;
  const size_t quads = pix * (size_t)(p.C >> 2);
  switch (op) {
    case YL_OP_POOL:
      hipLaunchKernelGGL(yl_pool_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, p);
      break;
    case YL_OP_COPY:
      hipLaunchKernelGGL(yl_copy_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, p);
      break;
    case YL_OP_LN:
      hipLaunchKernelGGL(yl_ln_kernel, dim3((unsigned)((pix + 3) / 4)), dim3(256), 0, st, p);
      break;
    case YL_OP_GRN: {
      const int cq = p.C >> 2;
      hipLaunchKernelGGL(yl_grn_sumsq_kernel, dim3((unsigned)((cq + 255) / 256), (unsigned)p.P, (unsigned)p.B), dim3(256), 0, st, p);
      hipLaunchKernelGGL(yl_grn_gate_kernel, dim3((unsigned)p.B), dim3(1024), 0, st, p);
      break;
    }
    case YL_OP_ACTPASS:
      hipLaunchKernelGGL(yl_act_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, p);
      break;
    case YL_OP_NHWC4:
      hipLaunchKernelGGL(yl_nhwc4_kernel, dim3((unsigned)(((size_t)p.B * p.H * p.W + 255) / 256)), dim3(256), 0, st, p);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
