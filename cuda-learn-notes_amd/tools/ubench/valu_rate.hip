// Issue rate of the VALU instructions the attention softmax is made of (gfx950): cycles per wave64 instruction, measured with
// s_memtime around an unrolled chain of INDEPENDENT instructions (8 accumulators), one wave per SIMD and two waves per SIMD.
// Backs the statements in DESIGN.md section 4.2 / 9 about the exponentials (v_exp_f32 vs full-rate and packed instructions).
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(512) void rate_k(float* out, long long* cyc, int iters) {
  float a[8];
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = -1.0f - 0.001f * (threadIdx.x + i), p[i] = f2{a[i], a[i] * 0.5f};
  const float c = 0.999f;
  const f2 pc = f2{0.999f, 1.001f};
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define PKF(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
#define PKA(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define CVT(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define MX3(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define LSA(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(c));
#define E16(i) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 0) { REP8(EXP) }
      if constexpr (OP == 1) { REP8(ADD) }
      if constexpr (OP == 2) { REP8(FMA) }
      if constexpr (OP == 3) { REP8(PKF) }
      if constexpr (OP == 4) { REP8(PKA) }
      if constexpr (OP == 5) { REP8(CVT) }
      if constexpr (OP == 6) { REP8(MX3) }
      if constexpr (OP == 7) { REP8(RCP) }
      if constexpr (OP == 8) { REP8(LSA) }
      if constexpr (OP == 9) { REP8(E16) }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rate_k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h = 0;
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 64.0 * iters;  // instructions per wave
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx9: report the event time and derive cycles at the measured sclk instead
    printf("VR %-18s %d wave(s)/SIMD: %8.3f us for %d instr per wave -> %6.2f ns per instr per SIMD (x sclk GHz = cycles), counter ticks %lld\n", name,
           waves_per_simd, ms * 1e3, (int)n, ms * 1e6 / (n * waves_per_simd), h);
  }
}

int main() {
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, 512 * 256 * 4), (void)hipMalloc(&cyc, 8);
  run<1>("v_add_f32", out, cyc);
  run<2>("v_fma_f32", out, cyc);
  run<3>("v_pk_fma_f32", out, cyc);
  run<4>("v_pk_add_f32", out, cyc);
  run<0>("v_exp_f32", out, cyc);
  run<9>("v_exp_f16", out, cyc);
  run<7>("v_rcp_f32", out, cyc);
  run<5>("v_cvt_pk_f16_f32", out, cyc);
  run<6>("v_max3_f32", out, cyc);
  run<8>("v_lshl_add_u32", out, cyc);
  return 0;
}
