// Does the accumulator register class decide whether a wave's MFMA stream overlaps with its SIMD partner's VALU
// stream? (gfx950, 512-thread workgroups: waves w and w+4 share a SIMD.) Group A (waves 0-3) runs a chain of
// v_mfma_f32_32x32x16_f16 over 4 accumulators held either in VGPRs ("+v") or in AGPRs ("+a"); group B (waves 4-7)
// runs independent v_fma_f32 / v_exp_f32 / a softmax-like mix. Prints wall time per combination.
//   hipcc --offload-arch=gfx950 -O3 overlap2.hip -o overlap2 && ./overlap2
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int ACC>
__device__ __forceinline__ void run_mfma(float* out, int n) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (threadIdx.x + i)), b[i] = (_Float16)(0.02f * (i + 1));
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < n; ++it) {
    if constexpr (ACC == 0) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
    } else {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c2) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c3) : "v"(a), "v"(b));
    }
  }
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int KIND>
__device__ __forceinline__ void run_valu(float* out, int n) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = 1.0f + i + threadIdx.x * 1e-3f;
  float b = 0.999f, c = 0.25f;
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if constexpr (KIND == 4) {  // softmax-like: fma, exp, add, max3 (+ a cvt_pk every other element)
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (i & 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        else asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[threadIdx.x] = s;
}
template <int ACC, int GA, int GB, int PRIO = 0>
__global__ __launch_bounds__(512) void k(float* out, int nm, int nv) {
  const int grp = threadIdx.x >> 8;
  float* o = out + blockIdx.x * 512;
  // PRIO 1: the VALU group raises its priority; PRIO 2: the MFMA group raises its priority
  if constexpr (PRIO == 1) { if ((grp == 0 ? GA : GB) >= 2) __builtin_amdgcn_s_setprio(3); }
  if constexpr (PRIO == 2) { if ((grp == 0 ? GA : GB) == 1) __builtin_amdgcn_s_setprio(3); }
  if (grp == 0) {
    if constexpr (GA == 1) run_mfma<ACC>(o, nm);
    if constexpr (GA >= 2) run_valu<GA>(o, nv);
  } else {
    if constexpr (GB == 1) run_mfma<ACC>(o, nm);
    if constexpr (GB >= 2) run_valu<GB>(o, nv);
  }
}
template <int ACC, int GA, int GB, int PRIO = 0>
void run(const char* tag, int nm, int nv) {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<ACC, GA, GB, PRIO>), dim3(256), dim3(512), 0, 0, out, nm, nv);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<ACC, GA, GB, PRIO>), dim3(256), dim3(512), 0, 0, out, nm, nv);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %8.1f us\n", tag, ms * 1e3 / 10);
  hipFree(out);
}
int main() {
  const int nm = 2500;  // x4 = 10000 MFMAs = 320k pipe cycles
  for (int pass = 0; pass < 2; ++pass) {
    run<0, 1, 0>("A: MFMA acc=VGPR   B: idle", nm, 0);
    run<1, 1, 0>("A: MFMA acc=AGPR   B: idle", nm, 0);
    run<0, 0, 2>("A: idle            B: fma x80000", nm, 5000);
    run<0, 2, 2>("A: fma x80000      B: fma x80000", nm, 5000);
    run<0, 0, 3>("A: idle            B: exp x32000", nm, 2000);
    run<0, 0, 4>("A: idle            B: softmax-mix x16000x4", nm, 1000);
    run<0, 1, 2>("A: MFMA acc=VGPR   B: fma x80000", nm, 5000);
    run<1, 1, 2>("A: MFMA acc=AGPR   B: fma x80000", nm, 5000);
    run<0, 1, 3>("A: MFMA acc=VGPR   B: exp x32000", nm, 2000);
    run<1, 1, 3>("A: MFMA acc=AGPR   B: exp x32000", nm, 2000);
    run<0, 1, 4>("A: MFMA acc=VGPR   B: softmax-mix", nm, 1000);
    run<1, 1, 4>("A: MFMA acc=AGPR   B: softmax-mix", nm, 1000);
    run<0, 2, 1>("A: fma x80000      B: MFMA (younger)", nm, 5000);
    run<0, 4, 1>("A: softmax-mix     B: MFMA (younger)", nm, 1000);
    run<0, 1, 2, 1>("A: MFMA            B: fma, fma wave at prio 3", nm, 5000);
    run<0, 1, 4, 1>("A: MFMA            B: softmax-mix at prio 3", nm, 1000);
    run<0, 1, 3, 1>("A: MFMA            B: exp at prio 3", nm, 2000);
    run<0, 1, 2, 2>("A: MFMA at prio 3  B: fma", nm, 5000);
    run<0, 2, 1, 1>("A: fma at prio 3   B: MFMA (younger)", nm, 5000);
    run<0, 1, 1>("A: MFMA acc=VGPR   B: MFMA acc=VGPR", nm, 0);
    run<1, 1, 1>("A: MFMA acc=AGPR   B: MFMA acc=AGPR", nm, 0);
  }
  return 0;
}
