// How many VALU issue cycles does one v_mfma_f32_16x16x32_f16 take away from the SIMD's other work (gfx950)?
// The attention kernels' issue model prices an MFMA at ONE 4-clock issue slot of the VALU port and
// predicts 2590 clocks per tile pair where the D = 64 kernel measures 4320. This sweep measures the price directly:
// a loop of { 1 MFMA ; k x OP } for k = 0 .. 8, one and two waves per SIMD, for
//   OP   = v_add_f32 (VOP2) | v_fma_f32 (VOP3) | v_exp_f32 (transcendental) | v_cvt_pk_f16_f32
//   FORM = 0: C in VGPRs, 8 independent accumulators      1: C in AGPRs
//          2: C = inline 0 (chain start: no C read)        3: A/B/C all fresh registers per MFMA (no operand reuse)
//          4: v_mfma_f32_32x32x16_f16, the unit is { 1 MFMA ; 2k x OP } (same flops per OP)
//          5: no MFMA at all (the OP stream alone)
// If the MFMA costs P port clocks and OP costs c, time(k) = max(16, P + k c) per unit and SIMD-wave: the knee of the
// curve gives P. Event-timed; prints ns per unit per SIMD and the same in shader clocks (s_memtime delta of wave 0).
//   hipcc --offload-arch=gfx950 -O3 mfma_port.hip -o mfma_port && ./mfma_port
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int OP>
__device__ __forceinline__ void op(float& x, float y, float z) {
  if constexpr (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  if constexpr (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
  if constexpr (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if constexpr (OP == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  if constexpr (OP == 4) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(y));
  if constexpr (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
}

template <int FORM, int OP, int K>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* clk, int n) {
  h8 a[8], b[8];
  for (int j = 0; j < 8; ++j)
    for (int i = 0; i < 8; ++i) a[j][i] = (_Float16)(0.001f * ((threadIdx.x & 15) + i + j)), b[j][i] = (_Float16)(0.002f * (i + 1 + j));
  f4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
  f16v d[2];
  for (int i = 0; i < 2; ++i)
    for (int e = 0; e < 16; ++e) d[i][e] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = -1.0f - 0.125f * i;
  float y = 0.999f, z = -0.25f;
  asm volatile("" : "+v"(y), "+v"(z));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if constexpr (FORM == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[r]) : "v"(a[0]), "v"(b[0]));
      if constexpr (FORM == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[r]) : "v"(a[0]), "v"(b[0]));
      if constexpr (FORM == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(c[r]) : "v"(a[0]), "v"(b[0]));
      if constexpr (FORM == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[r]) : "v"(a[r]), "v"(b[(r + 3) & 7]));
      if constexpr (FORM == 4) {
        if ((r & 1) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d[(r >> 1) & 1]) : "v"(a[0]), "v"(b[0]));
      }
#pragma unroll
      for (int i = 0; i < K; ++i) op<OP>(x[(r * K + i) & 7], y, z);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][3] + x[i];
  for (int i = 0; i < 2; ++i) r += d[i][0] + d[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

static const char* OPN[] = {"v_add_f32", "v_fma_f32", "v_exp_f32", "v_cvt_pk_f16_f32", "v_pk_add_f16", "v_max3_f32"};
static const char* FN[] = {"16x16x32 C=VGPR", "16x16x32 C=AGPR", "16x16x32 C=0", "16x16x32 fresh A/B", "32x32x16 (per half)", "no MFMA"};

template <int FORM, int OP, int K>
void run1(float* out, unsigned long long* clk, float* res_ns, float* res_clk) {
  const int n = 2000;
  for (int wps = 1; wps <= 2; ++wps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FORM, OP, K>), dim3(256), dim3(256 * wps), 0, 0, out, clk, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<FORM, OP, K>), dim3(256), dim3(256 * wps), 0, 0, out, clk, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    res_ns[wps - 1] = ms * 1e6f / (n * 8.0f * wps);
    res_clk[wps - 1] = (float)c / (n * 8.0f * wps);  // s_memtime ticks (100 MHz on gfx950? printed raw) per unit per SIMD
    (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
  }
}

template <int FORM, int OP>
void sweep(float* out, unsigned long long* clk) {
  float ns[9][2], ck[9][2];
  run1<FORM, OP, 0>(out, clk, ns[0], ck[0]);
  run1<FORM, OP, 1>(out, clk, ns[1], ck[1]);
  run1<FORM, OP, 2>(out, clk, ns[2], ck[2]);
  run1<FORM, OP, 3>(out, clk, ns[3], ck[3]);
  run1<FORM, OP, 4>(out, clk, ns[4], ck[4]);
  run1<FORM, OP, 5>(out, clk, ns[5], ck[5]);
  run1<FORM, OP, 6>(out, clk, ns[6], ck[6]);
  run1<FORM, OP, 8>(out, clk, ns[7], ck[7]);
  static const int KS[] = {0, 1, 2, 3, 4, 5, 6, 8};
  for (int w = 0; w < 2; ++w) {
    printf("MP %-20s + k x %-17s %d wave/SIMD ns/unit:", FN[FORM], OPN[OP], w + 1);
    for (int i = 0; i < 8; ++i) printf(" k=%d %6.2f", KS[i], ns[i][w]);
    printf("\n");
    printf("MP %-20s + k x %-17s %d wave/SIMD clk/unit:", FN[FORM], OPN[OP], w + 1);
    for (int i = 0; i < 8; ++i) printf(" k=%d %6.2f", KS[i], ck[i][w]);
    printf("\n");
  }
}

int main() {
  float* out;
  unsigned long long* clk;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&clk, 64);
  for (int pass = 0; pass < 2; ++pass) {
    sweep<5, 0>(out, clk), sweep<5, 1>(out, clk), sweep<5, 2>(out, clk), sweep<5, 3>(out, clk), sweep<5, 5>(out, clk);
    sweep<0, 0>(out, clk), sweep<0, 1>(out, clk), sweep<0, 2>(out, clk), sweep<0, 3>(out, clk), sweep<0, 5>(out, clk);
    sweep<1, 1>(out, clk), sweep<1, 2>(out, clk);
    sweep<2, 0>(out, clk), sweep<2, 1>(out, clk), sweep<2, 2>(out, clk);
    sweep<3, 1>(out, clk), sweep<3, 2>(out, clk);
    sweep<4, 0>(out, clk), sweep<4, 1>(out, clk), sweep<4, 2>(out, clk);
  }
  return 0;
}
