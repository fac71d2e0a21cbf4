// Pure ping-pong: can a wave that issues ONLY softmax arithmetic (no MFMA in its stream) run in the shadow of its SIMD partner's
// back-to-back MFMAs? (gfx950; the question behind VERDICT r5 #1.) tools/ubench/mfma_port.hip shows that two waves running the
// SAME mixed loop { MFMA ; k VALU } add their matrix and VALU times; tools/ubench/overlap2.hip (round 1) showed a pure-VALU wave
// at raised priority hiding most of a pure-MFMA wave. The shipped D = 64 attention kernel's phases are both MIXED (phase A:
// QK^T MFMAs + exponentials, phase B: PV MFMAs + deferred exponentials). This loop has the kernel's per-tile work
//   X = 64 v_mfma_f32_16x16x32_f16 (32 as 16 two-step chains from a splat = QK^T of a 128-key x 32-row tile at D = 64,
//       32 into 8 running accumulators = PV), optionally with the 32 ds_read_b128-equivalents of K / V fragments
//   Y = 64 v_exp_f32 + 64 v_add_f32 + 32 v_cvt_pk_f16_f32 + 24 v_max3_f32 + 16 v_fma_f32 (the 199-instruction softmax slice)
// per wave and period, 8 waves = 2 per SIMD, in these arrangements:
//   MODE 0  X only (both groups)                 MODE 1  Y only (both groups)
//   MODE 2  mixed: every wave runs [X half with Y half interleaved per 2 MFMAs] twice, one barrier per half (the shipped shape)
//   MODE 3  pure ping-pong: group 0 runs X while group 1 runs Y, barrier, swap, barrier
//   MODE 4  as 3, s_setprio 1 in Y / 0 in X       MODE 5  as 3, s_setprio 1 in X / 0 in Y
//   MODE 6  as 4 without the workgroup barriers (free-running)
//   MODE 7  as 2, but the block's 8 exponentials issued ONE behind each of the 8 MFMAs of the wave's own stream (round 6, late)
// Prints shader clocks per period per SIMD (two wave-tiles) and ns. The shipped kernel: 4320 clocks per tile pair.
//   hipcc --offload-arch=gfx950 -O3 pingpong.hip -o pingpong && ./pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define SB __builtin_amdgcn_sched_barrier(0)

struct St {
  h8 q[2][2];    // Q fragments (2 query blocks x 2 k-steps)
  f4 s[8][2];    // S^T of the tile: 8 key blocks x 2 query blocks
  h8 p[4][2];    // P^T: 4 k-steps of 32 keys x 2 query blocks
  f4 o[4][2];    // O^T: 4 d blocks x 2 query blocks
  f4 minit[2];
  float psum[2];
  h8 kf, vf;      // stand-ins of the K / V fragments in the registers-only form (opaque per use: no MFMA is merged, no copy is emitted)
};

template <bool LDS>
__device__ __forceinline__ void phase_x(St& st, const char* smem, int lane, int half, int nhalf) {
  // QK^T of key blocks [half * 8 / nhalf, ...) and PV k-steps of the same share
  const int kb0 = half * (8 / nhalf), kb1 = kb0 + 8 / nhalf;
  h8 kf, vf;
#pragma unroll
  for (int kb = kb0; kb < kb1; ++kb) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (LDS) kf = *reinterpret_cast<const h8*>(smem + ((kb * 2 + ks) * 1024 + lane * 16));
      else { asm volatile("" : "+v"(st.kf)); kf = st.kf; }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        st.s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, st.q[qb][ks], ks == 0 ? st.minit[qb] : st.s[kb][qb], 0, 0, 0);
        asm volatile("" : "+v"(st.s[kb][qb]) : "v"(kf), "v"(st.q[qb][ks]));
      }
    }
  }
  const int u0 = half * (4 / nhalf), u1 = u0 + 4 / nhalf;
#pragma unroll
  for (int u = u0; u < u1; ++u) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if constexpr (LDS) vf = *reinterpret_cast<const h8*>(smem + 16384 + ((u * 4 + b) * 1024 + lane * 16));
      else { asm volatile("" : "+v"(st.vf)); vf = st.vf; }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        st.o[b][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, st.p[u][qb], st.o[b][qb], 0, 0, 0);
        asm volatile("" : "+v"(st.o[b][qb]) : "v"(vf), "v"(st.p[u][qb]));
      }
    }
  }
}

template <bool LDS>
__device__ __forceinline__ void phase_x_block(St& st, const char* smem, int lane, int kb) {  // the 8 MFMAs that belong to one 16-key block
  h8 kf, vf;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    if constexpr (LDS) kf = *reinterpret_cast<const h8*>(smem + ((kb * 2 + ks) * 1024 + lane * 16));
    else { asm volatile("" : "+v"(st.kf)); kf = st.kf; }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      st.s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, st.q[qb][ks], ks == 0 ? st.minit[qb] : st.s[kb][qb], 0, 0, 0);
      asm volatile("" : "+v"(st.s[kb][qb]) : "v"(kf), "v"(st.q[qb][ks]));
    }
  }
  const int u = kb >> 1;
#pragma unroll
  for (int b = (kb & 1) * 2; b < (kb & 1) * 2 + 2; ++b) {
    if constexpr (LDS) vf = *reinterpret_cast<const h8*>(smem + 16384 + ((u * 4 + b) * 1024 + lane * 16));
    else { asm volatile("" : "+v"(st.vf)); vf = st.vf; }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      st.o[b][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, st.p[u][qb], st.o[b][qb], 0, 0, 0);
      asm volatile("" : "+v"(st.o[b][qb]) : "v"(vf), "v"(st.p[u][qb]));
    }
  }
}

__device__ __forceinline__ void y_block(St& st, int kb) {  // one 16-key block: 8 exp, 8 add, 4 cvt, 3 max3, 2 fma
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float mx = st.s[kb][qb][0];
    mx = fmaxf(fmaxf(mx, st.s[kb][qb][1]), st.s[kb][qb][2]);
    if (qb == 0) mx = fmaxf(fmaxf(mx, st.s[kb][qb][3]), st.s[kb][1][0]);
    asm volatile("" ::"v"(mx));
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const float a0 = __builtin_amdgcn_exp2f(st.s[kb][qb][r]), a1 = __builtin_amdgcn_exp2f(st.s[kb][qb][r + 1]);
      st.psum[qb] += a0 + a1;
      const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
      asm volatile("" ::"v"(a), "v"(st.psum[qb]));
      const int u = kb >> 1, e = (kb & 1) * 4 + r;
      st.p[u][qb][e] = a[0], st.p[u][qb][e + 1] = a[1];
    }
    st.minit[qb][kb & 3] = __builtin_fmaf(st.minit[qb][kb & 3], 0.999f, mx * 1e-9f);
  }
}
// MODE 7: the 8 MFMAs of key block kbx with the 8 exponentials of key block kby ONE behind each MFMA (tools/ubench/mfma_port.hip: a single v_exp_f32 behind
// an MFMA of the same wave costs 0.3 of its price, a second one nearly all of it), the block's other softmax items (adds, conversions, max3, fma) behind them.
template <bool LDS>
__device__ __forceinline__ void xy_block_interleaved(St& st, const char* smem, int lane, int kbx, int kby) {
  h8 kf, vf;
  float e[2][4];
  int n = 0;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    if constexpr (LDS) kf = *reinterpret_cast<const h8*>(smem + ((kbx * 2 + ks) * 1024 + lane * 16));
    else { asm volatile("" : "+v"(st.kf)); kf = st.kf; }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      st.s[kbx][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, st.q[qb][ks], ks == 0 ? st.minit[qb] : st.s[kbx][qb], 0, 0, 0);
      asm volatile("" : "+v"(st.s[kbx][qb]) : "v"(kf), "v"(st.q[qb][ks]));
      SB;
      e[n >> 2][n & 3] = __builtin_amdgcn_exp2f(st.s[kby][n >> 2][n & 3]);
      asm volatile("" : "+v"(e[n >> 2][n & 3]));
      SB;
      ++n;
    }
  }
  const int u = kbx >> 1;
#pragma unroll
  for (int b = (kbx & 1) * 2; b < (kbx & 1) * 2 + 2; ++b) {
    if constexpr (LDS) vf = *reinterpret_cast<const h8*>(smem + 16384 + ((u * 4 + b) * 1024 + lane * 16));
    else { asm volatile("" : "+v"(st.vf)); vf = st.vf; }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      st.o[b][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, st.p[u][qb], st.o[b][qb], 0, 0, 0);
      asm volatile("" : "+v"(st.o[b][qb]) : "v"(vf), "v"(st.p[u][qb]));
      SB;
      e[n >> 2][n & 3] = __builtin_amdgcn_exp2f(st.s[kby][n >> 2][n & 3]);
      asm volatile("" : "+v"(e[n >> 2][n & 3]));
      SB;
      ++n;
    }
  }
  // the rest of y_block(kby) on the exponentials taken above
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float mx = st.s[kby][qb][0];
    mx = fmaxf(fmaxf(mx, st.s[kby][qb][1]), st.s[kby][qb][2]);
    if (qb == 0) mx = fmaxf(fmaxf(mx, st.s[kby][qb][3]), st.s[kby][1][0]);
    asm volatile("" ::"v"(mx));
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const float a0 = e[qb][r], a1 = e[qb][r + 1];
      st.psum[qb] += a0 + a1;
      const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
      asm volatile("" ::"v"(a), "v"(st.psum[qb]));
      const int uu = kby >> 1, ee = (kby & 1) * 4 + r;
      st.p[uu][qb][ee] = a[0], st.p[uu][qb][ee + 1] = a[1];
    }
    st.minit[qb][kby & 3] = __builtin_fmaf(st.minit[qb][kby & 3], 0.999f, mx * 1e-9f);
  }
}
__device__ __forceinline__ void phase_y(St& st, int half, int nhalf) {
#pragma unroll
  for (int kb = half * (8 / nhalf); kb < (half + 1) * (8 / nhalf); ++kb) y_block(st, kb);
}

template <int MODE, bool LDS>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* clk, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 8;
  if (LDS) {
    for (int i = threadIdx.x; i < 32768 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
  }
  St st;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 8; ++e) st.q[a][b][e] = (_Float16)(0.01f * ((lane & 15) + e + a + 2 * b));
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) st.s[a][b] = f4{-1.f, -2.f, -0.5f, -3.f};
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      st.o[a][b] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 8; ++e) st.p[a][b][e] = (_Float16)0.01f;
    }
  st.minit[0] = st.minit[1] = f4{-4.f, -4.f, -4.f, -4.f};
  st.psum[0] = st.psum[1] = 0.f;
  st.kf = st.q[0][0], st.vf = st.q[1][1];
  asm volatile("" : "+v"(st.minit[0]), "+v"(st.minit[1]));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (MODE >= 3 && MODE != 7) {
    if (grp == 1) {
      if constexpr (MODE == 6) phase_y(st, 0, 1);
      else __builtin_amdgcn_s_barrier();
    }
  }
  for (int it = 0; it < n; ++it) {
    if constexpr (MODE == 0) {
      phase_x<LDS>(st, smem, lane, 0, 1);
      SB;
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int a = 0; a < 8; ++a) asm volatile("" : "+v"(st.s[a][0]), "+v"(st.s[a][1]));
      phase_y(st, 0, 1);
      SB;
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
    } else if constexpr (MODE == 2) {
      // the shipped shape: two mixed halves; per key block 4 QK^T MFMAs, the previous block's softmax items, 4 PV MFMAs
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int kb = h * 4; kb < h * 4 + 4; ++kb) {
          phase_x_block<LDS>(st, smem, lane, kb);
          SB;
          y_block(st, (kb + 7) & 7);
          SB;
        }
        __builtin_amdgcn_s_barrier();
      }
    } else if constexpr (MODE == 7) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int kb = h * 4; kb < h * 4 + 4; ++kb) {
          xy_block_interleaved<LDS>(st, smem, lane, kb, (kb + 7) & 7);
          SB;
        }
        __builtin_amdgcn_s_barrier();
      }
    } else {
      // every wave runs [X, barrier, Y, barrier]; group 1 entered the loop one barrier late, so its X meets group 0's Y
      if constexpr (MODE == 4 || MODE == 6) __builtin_amdgcn_s_setprio(0);
      if constexpr (MODE == 5) __builtin_amdgcn_s_setprio(1);
      SB;
      phase_x<LDS>(st, smem, lane, 0, 1);
      SB;
      if constexpr (MODE != 6) __builtin_amdgcn_s_barrier();
      if constexpr (MODE == 4 || MODE == 6) __builtin_amdgcn_s_setprio(1);
      if constexpr (MODE == 5) __builtin_amdgcn_s_setprio(0);
      SB;
      phase_y(st, 0, 1);
      SB;
      if constexpr (MODE != 6) __builtin_amdgcn_s_barrier();
    }
  }
  if constexpr (MODE >= 3 && MODE != 6 && MODE != 7) {
    if (grp == 0) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = st.psum[0] + st.psum[1];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) r += st.o[a][b][0] + st.o[a][b][3] + (float)st.p[a][b][0];
#pragma unroll
  for (int a = 0; a < 8; ++a) r += st.s[a][0][1] + st.s[a][1][2];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE, bool LDS>
void run(const char* tag, float* out, unsigned long long* clk) {
  const int n = 2000;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, LDS>), dim3(256), dim3(512), 65536, 0, out, clk, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, LDS>), dim3(256), dim3(512), 65536, 0, out, clk, n);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c = 0;
  (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("PP %-74s %s: %8.1f clk per period (wave 0), %7.1f ns per period = %6.0f TF-equivalent at D = 64\n", tag, LDS ? "with LDS fragment reads" : "registers only        ",
         (double)c / n, ms * 1e6 / n, 256.0 * 8 * 64 * 16384 / (ms * 1e-3 / n) * 1e-12);
  (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
}

int main() {
  float* out;
  unsigned long long* clk;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&clk, 64);
  for (int pass = 0; pass < 2; ++pass) {
    run<0, false>("X only: 64 MFMAs per wave and period", out, clk);
    run<1, false>("Y only: the softmax slice", out, clk);
    run<2, false>("mixed halves (the shipped shape), barrier per half", out, clk);
    run<7, false>("mixed halves, ONE exponential behind each MFMA (1 : 1), rest of the slice per block", out, clk);
    run<3, false>("pure ping-pong: X in one group while Y in the other, 2 barriers", out, clk);
    run<4, false>("pure ping-pong, priority 1 in Y", out, clk);
    run<5, false>("pure ping-pong, priority 1 in X", out, clk);
    run<6, false>("pure ping-pong, priority 1 in Y, no barriers", out, clk);
    run<0, true>("X only: 64 MFMAs per wave and period", out, clk);
    run<2, true>("mixed halves (the shipped shape), barrier per half", out, clk);
    run<7, true>("mixed halves, ONE exponential behind each MFMA (1 : 1), rest of the slice per block", out, clk);
    run<3, true>("pure ping-pong: X in one group while Y in the other, 2 barriers", out, clk);
    run<4, true>("pure ping-pong, priority 1 in Y", out, clk);
    run<6, true>("pure ping-pong, priority 1 in Y, no barriers", out, clk);
  }
  return 0;
}
