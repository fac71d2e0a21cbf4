// Same-wave interleave capacity (gfx950): each wave runs   loop { v_mfma_f32_32x32x16_f16 ; K x VALU }   with the
// MFMAs rotating over 4 accumulators and the VALU ops independent of them. If VALU issued in the shadow of the
// wave's own MFMA is free, time stays at the MFMA-only figure until K exceeds the number of hidden slots.
// Run with 1 and 2 waves per SIMD (256 / 512 threads).   hipcc --offload-arch=gfx950 -O3 -w interleave.hip -o interleave
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int K, int KIND>
__global__ __launch_bounds__(512) void k(float* out, int n) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (threadIdx.x + i)), b[i] = (_Float16)(0.02f * (i + 1));
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + i + threadIdx.x * 1e-3f;
  const float p = 0.999f, q = 0.25f;
#define FILL(J)                                                                                                   \
  _Pragma("unroll") for (int e = 0; e < K; ++e) {                                                                 \
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(J * K + e) & 7]) : "v"(p), "v"(q)); \
    if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(J * K + e) & 7]));                         \
    if constexpr (KIND == 2) {                                                                                    \
      if (((J * K + e) & 3) == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(J * K + e) & 7]));                    \
      else if (((J * K + e) & 3) == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[(J * K + e) & 7]) : "v"(p)); \
      else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(J * K + e) & 7]) : "v"(p), "v"(q));                  \
    }                                                                                                             \
  }
  for (int it = 0; it < n; ++it) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
    FILL(0)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
    FILL(1)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
    FILL(2)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
    FILL(3)
  }
  float s = c0[0] + c1[1] + c2[2] + c3[3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int K, int KIND>
void run(int threads) {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int n = 2500;
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<K, KIND>), dim3(256), dim3(threads), 0, 0, out, n);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<K, KIND>), dim3(256), dim3(threads), 0, 0, out, n);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const char* kn[] = {"fma", "exp", "mix(fma,exp,fma,cvt_pk)"};
  // 10000 MFMAs per wave; waves per SIMD = threads / 256
  printf("%-24s K=%2d  %d wave/SIMD  %8.1f us   = %6.1f ns per MFMA slot per SIMD\n", kn[KIND], K, threads / 256,
         ms * 1e3 / 10, ms * 1e6 / 10 / (10000.0 * (threads / 256)));
  hipFree(out);
}
template <int KIND>
void sweep() {
  for (int t : {256, 512}) {
    run<0, KIND>(t); run<1, KIND>(t); run<2, KIND>(t); run<3, KIND>(t); run<4, KIND>(t);
    run<5, KIND>(t); run<6, KIND>(t); run<8, KIND>(t); run<12, KIND>(t);
  }
}
int main() {
  sweep<0>();
  sweep<1>();
  sweep<2>();
  return 0;
}
