// VALU issue-rate probe (gfx950): cycles per v_fma_f32 / v_pk_fma_f32 / v_pk_add_f32 / v_mfma_f32_16x16x4_f32 for one wave per SIMD
// and for two waves per SIMD (one issuing MFMAs, one VALU: do they add?).   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  f32x2 p0 = {a, a}, p1 = {a + 1, a}, p2 = {a + 2, a}, p3 = {a + 3, a}, q = {b, b};
  float s0 = a, s1 = a + 1, s2 = a + 2, s3 = a + 3;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  // MODE 0: scalar fma, 1: pk fma, 2: mfma, 3: waves 0-3 mfma + waves 4-7 pk fma, 4: waves 0-3 mfma + waves 4-7 scalar fma, 5: pk add
  const int role = MODE <= 2 || MODE == 5 ? MODE : (wave < 4 ? 2 : (MODE == 3 ? 1 : 0));
  __syncthreads();
  const long long t0 = wall_clock64();
  const long long c0s = clock64();
  for (int i = 0; i < iters; ++i) {
    if (role == 0) {
      REP16(asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(b));)
    } else if (role == 1) {
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
    } else if (role == 5) {
      REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
    } else {
      REP16(asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));)
    }
  }
  const long long c1s = clock64();
  const long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = c1s - c0s; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = t1 - t0; }
  out[blockIdx.x * 512 + threadIdx.x] = s0 + s1 + s2 + s3 + p0.x + p1.y + p2.x + p3.y + c0.x + c1.y + c2.z + c3.w;
}
template <int MODE> void run(const char* name, int threads) {
  float* out; long long* cyc; const int iters = 2000;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 2 * 8);
  hipMemset(cyc, 0, 256 * 8 * 2 * 8);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256 * 16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n = iters * 64.0;
  printf("%-44s wave0: %.2f clk/instr (%.2f wall ticks)", name, h[0] / n, h[1] / n);
  if (threads == 512) printf("   wave4: %.2f clk/instr (%.2f wall ticks)", h[8] / n, h[9] / n);
  printf("\n");
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("v_fma_f32, 1 wave/SIMD", 256);
  run<1>("v_pk_fma_f32, 1 wave/SIMD", 256);
  run<5>("v_pk_add_f32, 1 wave/SIMD", 256);
  run<2>("v_mfma_f32_16x16x4_f32, 1 wave/SIMD", 256);
  run<0>("v_fma_f32, 2 waves/SIMD", 512);
  run<1>("v_pk_fma_f32, 2 waves/SIMD", 512);
  run<2>("mfma, 2 waves/SIMD", 512);
  run<3>("waves 0-3 mfma | waves 4-7 v_pk_fma_f32", 512);
  run<4>("waves 0-3 mfma | waves 4-7 v_fma_f32", 512);
  return 0;
}
