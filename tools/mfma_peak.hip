// Practical peak of v_mfma_f32_16x16x4_f32 on the whole chip (tools/run_mfma_peak.sh): every wave issues a long run of
// independent MFMAs (NACC accumulators round-robin), W waves per SIMD.  Prints TFLOP/s and the implied clock, so that
// "fraction of the 157.3 TF peak" in bench.py can be read against what the matrix pipe sustains under load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_run(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s.x == 123.456f) out[threadIdx.x] = s.x + s.y + s.z + s.w;
}
template <int NACC>
static void run(int waves_per_simd, int iters, float* d) {
  const int blocks = 256 * waves_per_simd;       // 256 CUs x (4 waves = one per SIMD) x waves_per_simd
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_run<NACC><<<blocks, 256>>>(d, 100, 1.0f, 1.0f);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    mfma_run<NACC><<<blocks, 256>>>(d, iters, 1.0f, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = (double)waves_per_simd * iters * 4 * NACC;
    const double flops = mfmas_per_simd * 1024.0 * 2048.0;
    const double ghz = mfmas_per_simd * 32.0 / (ms * 1e-3) * 1e-9;     // 8 passes x 4 cycles per MFMA
    printf("{\"nacc\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"tflops\": %.2f, \"implied_ghz_at_32_cycles_per_mfma\": %.3f}\n",
           NACC, waves_per_simd, ms, flops / (ms * 1e-3) * 1e-12, ghz);
  }
}
int main(int argc, char** argv) {
  float* d = nullptr;
  hipMalloc(&d, 4096);
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  run<4>(1, iters, d);
  run<8>(1, iters, d);
  run<8>(2, iters, d);
  run<16>(2, iters / 2, d);
  run<8>(4, iters / 2, d);
  // long run: does the rate fall once the power / thermal limits engage?
  run<8>(2, iters * 16, d);
  return 0;
}
