// f32 MFMA issue rate: back-to-back independent v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32 vs 4x4x1 (16 blocks), one / two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  if constexpr (KIND == 0) {
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f4v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}
template <int KIND, int NACC>
void run(const char* name, int blocks_per_cu, float* out) {
  const int iters = 4000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<KIND, NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<KIND, NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop_per = KIND == 0 ? 4096.0 : 2048.0;
  const double flops = (double)grid * 4 * iters * 4 * NACC * flop_per;
  printf("F32RATE %-34s %d WG/CU  %8.3f ms  %7.2f TF  (%.3f of 157.3)\n", name, blocks_per_cu, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 157.3);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  run<0, 4>("32x32x2 x4 accumulators", 1, out);
  run<0, 8>("32x32x2 x8 accumulators", 1, out);
  run<0, 4>("32x32x2 x4 accumulators", 2, out);
  run<1, 8>("16x16x4 x8 accumulators", 1, out);
  run<1, 16>("16x16x4 x16 accumulators", 1, out);
  run<1, 8>("16x16x4 x8 accumulators", 2, out);
  run<0, 2>("32x32x2 x2 accumulators", 1, out);
  run<1, 4>("16x16x4 x4 accumulators", 1, out);
  return 0;
}
