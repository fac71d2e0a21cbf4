// Does one wave's MFMA stream overlap with its SIMD partner's VALU stream? (gfx950, 512-thread workgroups:
// waves w and w+4 share a SIMD.)  Modes per wave-group: M = 32x32x16 f16 MFMA chain (4 accumulators),
// V = independent v_fma_f32, T = v_exp_f32, 0 = idle. Prints wall time per variant.
//   hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap && ./overlap
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define ITER 4000

__device__ __forceinline__ void run_mfma(float* out) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (threadIdx.x + i)), b[i] = (_Float16)(0.02f * (i + 1));
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < ITER / 4; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int KIND>
__device__ __forceinline__ void run_valu(float* out, int n) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = 1.0f + i + threadIdx.x * 1e-3f;
  float b = 0.5f, c = 0.25f;
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      else asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[threadIdx.x] = s;
}
// ga / gb: what waves 0-3 / 4-7 do: 0 idle, 1 MFMA (ITER instrs = ITER*32 pipe cycles), 2 fma (nv x 16 instrs), 3 exp
template <int GA, int GB>
__global__ __launch_bounds__(512) void k(float* out, int nv) {
  const int grp = threadIdx.x >> 8;
  float* o = out + blockIdx.x * 512;
  if (grp == 0) {
    if constexpr (GA == 1) run_mfma(o);
    if constexpr (GA == 2) run_valu<0>(o, nv);
    if constexpr (GA == 3) run_valu<1>(o, nv);
  } else {
    if constexpr (GB == 1) run_mfma(o);
    if constexpr (GB == 2) run_valu<0>(o, nv);
    if constexpr (GB == 3) run_valu<1>(o, nv);
  }
}
template <int GA, int GB>
void run(const char* tag, int nv) {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<GA, GB>), dim3(256), dim3(512), 0, 0, out, nv);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<GA, GB>), dim3(256), dim3(512), 0, 0, out, nv);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %8.1f us\n", tag, ms * 1e3 / 5);
  hipFree(out);
}
int main() {
  const int nv = 2000;  // 32000 VALU instrs per wave
  run<1, 0>("A: MFMA (4000)      B: idle", nv);
  run<1, 1>("A: MFMA             B: MFMA", nv);
  run<0, 2>("A: idle             B: fma (32000)", nv);
  run<2, 2>("A: fma              B: fma", nv);
  run<0, 3>("A: idle             B: exp (32000)", nv);
  run<1, 2>("A: MFMA             B: fma", nv);
  run<1, 3>("A: MFMA             B: exp", nv);
  run<2, 3>("A: fma              B: exp", nv);
  return 0;
}
