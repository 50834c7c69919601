// FETCH_SIZE / WRITE_SIZE calibration on kernels whose byte counts are known (MI355X_MICROARCH.md: "FETCH_SIZE reports
// exactly 1/2 of the bytes of a wide coalesced streaming read; other access widths and WRITE_SIZE are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Independent of the product kernels: every kernel
// here streams ONE buffer exactly once, in the lane access patterns the network's kernels use:
//   calib_x4    16 B per lane, consecutive lanes consecutive (float4 NHWC reads / glds16)
//   calib_x1     4 B per lane, consecutive (stem block: one dword per patch pixel)
//   calib_x3s8  12 B per lane at a lane stride of 8 B (stem block: three taps of a stride-2 input row -- neighbouring
//               lanes overlap by 4 B; unique bytes = the buffer once)
//   calib_w4    16 B per lane stores (float4 NHWC epilogues)
//   calib_w4s   16 B per lane stores in 64-B segments at a 384-B pitch (the D-fragment epilogue of a 96-channel layer)
// Build + run + summarise: tools/fetch_calib.sh (on the GPU box, under rocprofv3 --pmc).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };

__global__ void calib_x4(const f4* __restrict__ p, size_t n4, float* out) {
  f4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}
__global__ void calib_x1(const float* __restrict__ p, size_t n, float* out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678f) out[0] = s;
}
__global__ void calib_x3s8(const float* __restrict__ p, size_t n, float* out) {     // lane i: floats [2i, 2i+3)
  float s = 0;
  const size_t items = (n - 4) / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (size_t)gridDim.x * blockDim.x) {
    const F3 v = *reinterpret_cast<const F3*>(p + 2 * i);
    s += v.a + v.b + v.c;
  }
  if (s == 12345.678f) out[0] = s;
}
__global__ void calib_w4(f4* __restrict__ p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (f4){1.f, 2.f, 3.f, (float)i};
}
__global__ void calib_w4s(float* __restrict__ p, size_t rows) {                    // rows of 96 floats; one wave-store = 16 rows x 64 B
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t t = wave; t < rows / 16; t += nw)
    for (int nt = 0; nt < 6; ++nt)
      *reinterpret_cast<f4*>(p + (t * 16 + (lane & 15)) * 96 + nt * 16 + 4 * (lane >> 4)) = (f4){1.f, 2.f, 3.f, (float)nt};
}

int main() {
  const size_t bytes = (size_t)768 << 20;                    // 768 MiB: beyond the 256 MiB Infinity Cache
  float *buf, *out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  const dim3 g(256 * 8), b(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_x4, g, b, 0, 0, (const f4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(calib_x1, g, b, 0, 0, buf, bytes / 4, out);
    hipLaunchKernelGGL(calib_x3s8, g, b, 0, 0, buf, bytes / 4, out);
    hipLaunchKernelGGL(calib_w4, g, b, 0, 0, (f4*)buf, bytes / 16);
    hipLaunchKernelGGL(calib_w4s, g, b, 0, 0, buf, bytes / 384);
  }
  hipDeviceSynchronize();
  printf("bytes %zu\n", bytes);
  return 0;
}
