// What hides in the shadow of v_mfma_f32_16x16x32_f16 (gfx950)? The attention kernels issue, per pair of MFMAs, a slice of softmax
// arithmetic (4 v_exp_f32, 4 v_add_f32, 2 v_cvt_pk_f16_f32); measured in the kernel, MFMA time and VALU time ADD
// (profiles/r03_fa_clock_power_ablation.log). This isolates the pattern: a loop of { 2 MFMAs ; 10 VALU } where the VALU slice
//   MODE 0: absent (MFMA stream alone)            MODE 1: alone (no MFMAs)
//   MODE 2: works on registers the MFMAs never touch
//   MODE 3: reads the accumulators the MFMAs wrote FOUR MFMAs earlier (the kernel's dependency) -- accumulators in VGPRs
//   MODE 4: as 2, but MFMA M V(5) M V(5) instead of M M V(10)
//   MODE 5 / 6: the same flops on ONE v_mfma_f32_32x32x16_f16 per unit: alone / with the slice behind it (unrelated registers)
//   MODE 7: 32x32x16, the slice reading the accumulator written two MFMAs earlier
// at one and two waves per SIMD. Event-timed, ns per loop iteration per SIMD.
//   hipcc --offload-arch=gfx950 -O3 shadow16.hip -o shadow16 && ./shadow16
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define MFMA(C) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(C) : "v"(a), "v"(b));
#define MFMA32(C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(C) : "v"(a), "v"(b));
#define SLICE5(X0, X1, X2, X3, S)                                  \
  asm volatile("v_exp_f32 %0, %1" : "=v"(e0) : "v"(X0));           \
  asm volatile("v_exp_f32 %0, %1" : "=v"(e1) : "v"(X1));           \
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(e0), "v"(e1)); \
  asm volatile("v_add_f32 %0, %0, %1" : "+v"(S) : "v"(t));         \
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(e0), "v"(e1)); \
  asm volatile("" ::"v"(p));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int n) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.001f * ((threadIdx.x & 15) + i)), b[i] = (_Float16)(0.002f * (i + 1));
  f4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
  f4 x[2];
  x[0] = f4{-1.f, -2.f, -3.f, -4.f}, x[1] = f4{-1.5f, -2.5f, -3.5f, -4.5f};
  asm volatile("" : "+v"(x[0]), "+v"(x[1]));
  float s0 = 0.f, s1 = 0.f, e0, e1, t;
  unsigned p;
  for (int it = 0; it < (MODE >= 5 ? 0 : n); ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // 4 x (2 MFMAs + slice) per iteration: accumulators 2r, 2r+1; the slice reads (2r + 4) % 8, i.e. written 4-5 MFMAs ago
      f4& cr0 = c[(2 * r + 4) % 8];
      f4& cr1 = c[(2 * r + 5) % 8];
      if constexpr (MODE != 1) MFMA(c[2 * r]);
      if constexpr (MODE == 4) { SLICE5(x[0][0], x[0][1], x[0][2], x[0][3], s0) }
      if constexpr (MODE != 1) MFMA(c[2 * r + 1]);
      if constexpr (MODE == 4) { SLICE5(x[1][0], x[1][1], x[1][2], x[1][3], s1) }
      if constexpr (MODE == 1 || MODE == 2) { SLICE5(x[0][0], x[0][1], x[0][2], x[0][3], s0) SLICE5(x[1][0], x[1][1], x[1][2], x[1][3], s1) }
      if constexpr (MODE == 3) { SLICE5(cr0[0], cr0[1], cr0[2], cr0[3], s0) SLICE5(cr1[0], cr1[1], cr1[2], cr1[3], s1) }
    }
  }
  float r = s0 + s1;
  for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][3];
  if constexpr (MODE >= 5) {
    f16v d[4];
    for (int i = 0; i < 4; ++i)
      for (int e = 0; e < 16; ++e) d[i][e] = 0.f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f16v& dr = d[(q + 2) % 4];
        MFMA32(d[q]);
        if constexpr (MODE == 6) { SLICE5(x[0][0], x[0][1], x[0][2], x[0][3], s0) SLICE5(x[1][0], x[1][1], x[1][2], x[1][3], s1) }
        if constexpr (MODE == 7) { SLICE5(dr[0], dr[1], dr[2], dr[3], s0) SLICE5(dr[4], dr[5], dr[6], dr[7], s1) }
      }
    }
    r += s0 + s1;
    for (int i = 0; i < 4; ++i) r += d[i][0] + d[i][15];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char* tag, float* out) {
  const int n = 4000;
  for (int wps = 1; wps <= 2; ++wps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("SH %-58s %d wave(s)/SIMD: %8.1f us -> %6.2f ns per {2 MFMA + slice} per SIMD\n", tag, wps, ms * 1e3, ms * 1e6 / (n * 4.0 * wps));
  }
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  for (int pass = 0; pass < 2; ++pass) {
    run<0>("2 MFMA 16x16x32 alone", out);
    run<1>("slice alone (4 exp, 4 add, 2 cvt_pk)", out);
    run<2>("M M + slice on unrelated registers", out);
    run<3>("M M + slice reading accumulators written 4 MFMAs ago", out);
    run<4>("M V5 M V5, unrelated registers", out);
    run<5>("1 MFMA 32x32x16 alone (same flops)", out);
    run<6>("1 MFMA 32x32x16 + slice on unrelated registers", out);
    run<7>("1 MFMA 32x32x16 + slice reading the accumulator of 2 MFMAs ago", out);
  }
  return 0;
}
