// VALU issue-cost microbenchmark (gfx950): cycles per wave-instruction for the ops the attention softmax uses.
// One workgroup of 256 threads per CU-slot, 1 or 2 waves per SIMD; each kernel runs ITER x 16 independent
// instructions of one kind per wave and reports (s_memtime delta) / instructions.
//   hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost && ./valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 2000
#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, float seed) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x * 1e-3f;
  float b = seed * 0.5f, c = seed * 0.25f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if constexpr (KIND == 2) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
      if constexpr (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (KIND == 6) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (KIND == 7) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (KIND == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if constexpr (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (KIND == 10) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// packed f32: operands are register pairs
template <int KIND>
__global__ __launch_bounds__(512) void kpk(float* out, unsigned long long* cyc, float seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f2{seed + i, seed - i};
  f2 b = {seed * 0.5f, seed}, c = {seed * 0.25f, 1.f};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if constexpr (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if constexpr (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  f2 s = {0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
void run(const char* name, F launch, int threads) {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch(out, cyc, threads);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(out, cyc, threads);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256];
  hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < 256; ++i) avg += h[i];
  avg /= 256;
  // s_memtime/readcyclecounter ticks at a constant 100 MHz on some parts: also report wall-time based figure
  const double insts = (double)ITER * 16;
  printf("%-18s threads/WG %4d : %.2f counter-ticks/inst/wave, wall %.3f ms -> %.2f ns/inst/wave\n", name, threads,
         avg / insts, ms, ms * 1e6 / insts);
  hipFree(out);
  hipFree(cyc);
}

#define RUN(NAME, KERN)                                                                                         \
  for (int th : {256, 512})                                                                                     \
    run(NAME, [](float* o, unsigned long long* c, int t) { hipLaunchKernelGGL(KERN, dim3(256), dim3(t), 0, 0, o, c, 1.0f); }, th);

int main() {
  RUN("v_fma_f32", k<0>)
  RUN("v_exp_f32", k<1>)
  RUN("v_exp_f16", k<2>)
  RUN("v_add_f32", k<3>)
  RUN("v_max3_f32", k<4>)
  RUN("v_cvt_pk_f16_f32", k<5>)
  RUN("v_dot2_f32_f16", k<6>)
  RUN("v_pk_fma_f16", k<7>)
  RUN("v_rcp_f32", k<8>)
  RUN("v_mul_f32", k<9>)
  RUN("v_log_f32", k<10>)
  RUN("v_pk_fma_f32", kpk<0>)
  RUN("v_pk_add_f32", kpk<1>)
  RUN("v_pk_mul_f32", kpk<2>)
  return 0;
}
