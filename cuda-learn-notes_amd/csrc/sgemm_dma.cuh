// SGEMM on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32, tiles fed by LDS-DMA (round 6).
//
// Reference rungs served: kernels/sgemm/sgemm_wmma_tf32_stage.cu (sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages[_dsmem]; the reference
// rounds to TF32 -- gfx950 has no TF32, the product here is exact f32 with f32 accumulation, i.e. strictly closer to the script's
// torch.matmul row than the reference's own kernel).
//
// Why a second matrix-core kernel: the register-staged 128x128x16 kernel of rounds 2-5 (sgemm.hip sgemm_mfma_kernel) sits at 0.82 of
// the 157 TF f32 matrix peak and 0.93x rocBLAS sgemm at 4096^3. It pays, per 128x128 tile and 16-deep step, 4 global loads + 10 LDS
// stores per thread through VGPRs, a barrier every 32 MFMAs per wave, and moves 2x the L2 bytes of a 256x256 tile. This kernel:
//   * block tile (WM*TM*32) x (WN*TN*32), 256x256 by default: 8 waves = two per SIMD, wave tile 128x64 (4 x 2 MFMA tiles, 128
//     accumulator registers), one workgroup per CU;
//   * K in BK-deep stages through a ring of S >= 3 LDS slots (3 for the large tiles; 8 for the 64x128 tile on grids of one workgroup per CU, where a stage
//     lasts 0.4 us -- less than one trip to memory) filled by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write);
//     ONE barrier per stage = per 128 MFMAs of a wave at BK = 32; the DMA of stage t+S-1 is issued right behind the barrier that frees
//     its slot, so it has S-1 whole stages (>= 16k clocks) to land;
//   * A image: rows of BK floats (128 B = one full line at BK = 32), 16-byte chunks XOR-swizzled on the SOURCE side so that the
//     ds_read_b128 of 8 consecutive rows covers all 64 banks; k is permuted inside a group of 8 (lane half kh takes k = 8m + 4kh + s in
//     step s) so ONE b128 read feeds four MFMA steps of an A tile -- the same permutation on B keeps the sum intact (f32 addition
//     order inside a k8 group changes, nothing else);
//   * B image: k-major rows of BN floats as they lie in memory; lanes 0..31 read 32 consecutive words of row k, lanes 32..63 of row
//     k+4, whose chunks sit 128 B further (source-side XOR of chunk bit 3) so that the two halves use different banks.
// Per wave and k8 group: TM b128 + 4 TN b32 LDS reads for 4 TM TN MFMAs of 64 clocks each -- the LDS and the vector ALU are idle
// > 90 % of the time, which is the point: under the package power cap the clock is what the remaining energy buys.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"

namespace sgemm_dma {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// LDS-DMA issued from asm (as hgemm_mfma.cuh glds16_asm): hipcc cannot prove that an in-flight DMA does not alias the fragment reads
// of another ring slot and would drain vmcnt(0) before the first ds_read of every stage. Source = SGPR base + per-lane byte offset,
// destination = M0 (wave-uniform) + lane * 16; ordered only by the counted waits below + the barrier.
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int WM, int WN, int TM, int TN, int BK, int S>
struct Geo {
  static constexpr int NW = WM * WN, THREADS = NW * 64;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static constexpr int RB = BK * 4;            // bytes per A row in LDS
  static constexpr int CPR = BK / 4;           // 16-byte chunks per A row
  static constexpr int RPG = 256 / RB;         // A rows per 256-byte bank span
  static constexpr int RPI = 64 / CPR;         // A rows per DMA instruction (1 KiB)
  static constexpr int A_BYTES = BM * RB, B_BYTES = BK * BN * 4, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_I = A_BYTES / 1024 / NW, B_I = B_BYTES / 1024 / NW;  // DMA instructions per wave and stage
  static constexpr int LDS = S * STAGE;
  static_assert(A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0, "whole DMA instructions per wave");
  static_assert((A_I + B_I) <= (BK / 4) * TM * TN, "one DMA piece per MFMA of the second half stage");
  static_assert(BK % 8 == 0 && (BN * 4) % 512 == 0 && TN % 2 == 0, "k8 groups; B rows of whole 512-byte spans; column blocks in XOR pairs");
};

template <int WM, int WN, int TM, int TN, int BK, int S, bool KSPLIT = false>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN == 4 && TM * TN <= 8) ? 2 : 1) void sgemm_dma_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                float* __restrict__ C, int M, int N, int K, int tiles_n,
                                                                int swizzle) {
  using G = Geo<WM, WN, TM, TN, BK, S>;
  static_assert(S >= 3 && BK >= 16, "at least three slots: read t, landed t+1, in flight t+2 .. t+S-1; fragment double buffers need >= 2 k8 groups");
  static_assert((S - 2) * (G::A_I + G::B_I) < 64, "vmcnt field");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: it addresses M0)
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l31 = lane & 31, kh = lane >> 5;

  int bid = blockIdx.x;
  if (swizzle) {  // bijective XCD remap (workgroup b runs on XCD b % 8): each XCD gets a contiguous run of tiles
    const int nblk = gridDim.x, xcd = bid & 7, local = bid >> 3, q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * G::BM, n0 = tn * G::BN;

  // ---- DMA sources: a scalar base that advances one stage at a time + per-lane byte offsets that never change; destinations wave-uniform
  const char* abase = reinterpret_cast<const char*>(A + (size_t)m0 * K);
  const char* bbase = reinterpret_cast<const char*>(B + n0);
  const size_t astep = (size_t)BK * 4, bstep = (size_t)BK * N * 4;
  unsigned avo[G::A_I], bvo[G::B_I];
#pragma unroll
  for (int q = 0; q < G::A_I; ++q) {
    const int row = (wave * G::A_I + q) * G::RPI + lane / G::CPR;     // row of the tile this lane's chunk belongs to
    const int pos = lane % G::CPR;                                    // chunk position in the LDS row
    const int chunk = pos ^ ((row / G::RPG) % G::CPR);                // source chunk that lands there
    avo[q] = ((unsigned)row * (unsigned)K + chunk * 4) * 4u;
  }
  constexpr int BCPR = G::BN / 4;  // chunks per B row
#pragma unroll
  for (int q = 0; q < G::B_I; ++q) {
    const int unit = (wave * G::B_I + q) * 64 + lane;                 // chunk index in the stage's B image
    const int krow = unit / BCPR, pos = unit % BCPR;
    const int chunk = pos ^ (((krow >> 2) & 1) << 3);
    bvo[q] = ((unsigned)krow * (unsigned)N + chunk * 4) * 4u;
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // KSPLIT (few tiles, long K): blockIdx.y = 0 / 1 takes the first / second half of the stages and ADDS its product to a C the launcher zeroed --
  // two commutative fp32 additions per element (0 + p + q = 0 + q + p): the result does not depend on which half arrives first
  int nt = K / BK;
  if constexpr (KSPLIT) {
    const int first = (nt + 1) / 2;
    if (blockIdx.y == 0) {
      nt = first;
    } else {
      abase += (size_t)first * astep;
      bbase += (size_t)first * bstep;
      nt -= first;
    }
  }
  int nxt = 0;  // stage the bases point at
  // one 1-KiB piece of a stage's A / B image (pieces 0 .. A_I-1: A; then B); behind the last one the bases move on to the next stage
  // (a stage past the end of K re-reads the last one: the request count per stage stays fixed)
  constexpr int NDMA = G::A_I + G::B_I;
  auto issue_piece = [&](int slot, int d) __attribute__((always_inline)) {
    if (d < G::A_I)
      dma16(abase, avo[d], lds0 + slot * G::STAGE + (wave * G::A_I + d) * 1024);
    else
      dma16(bbase, bvo[d - G::A_I], lds0 + slot * G::STAGE + G::A_BYTES + (wave * G::B_I + d - G::A_I) * 1024);
    if (d == NDMA - 1 && nxt + 1 < nt) {
      ++nxt;
      abase += astep;
      bbase += bstep;
    }
  };
  auto issue = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < NDMA; ++d) issue_piece(slot, d);
  };

  // ---- fragment read addresses: one register per (slot, k8 group) for A and per (slot, column-block parity) for B, pinned (an address
  // recomputed in the loop is a VALU instruction, and on gfx950 a VALU instruction is matrix-pipe time); everything else is an offset field
  constexpr int MG = BK / 8, NSTEP = BK / 2;
  unsigned ab[S][MG], bb_[S][2];
  {
    const int sw = (l31 / G::RPG) % G::CPR;  // the 32-row tile offsets and the wave offset are multiples of RPG * CPR rows
#pragma unroll
    for (int sl = 0; sl < S; ++sl) {
#pragma unroll
      for (int m = 0; m < MG; ++m) {
        ab[sl][m] = lds0 + sl * G::STAGE + (unsigned)((wm * TM * 32 + l31) * G::RB + (((2 * m + kh) ^ sw) << 4));
        asm volatile("" : "+v"(ab[sl][m]));
      }
      // B: row k = 8m + 4kh + s, column block j of this wave: lanes of the upper half read the block their XOR put them in
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        bb_[sl][p] = lds0 + sl * G::STAGE + (unsigned)(G::A_BYTES + (4 * kh) * G::BN * 4 + ((wn * TN * 32 + ((p ^ kh) << 5) + l31) << 2));
        asm volatile("" : "+v"(bb_[sl][p]));
      }
    }
  }

  f16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f4v a[2][TM];   // [k8 group parity]
  float b[2][TN];  // [step parity]
  auto load_a = [&](int par, unsigned base) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a[par][i] = lds_ld<f4v>(base + i * 32 * G::RB);
  };
  auto load_b = [&](int par, const unsigned* base2, int krow) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < TN; ++j) b[par][j] = lds_ld<float>(base2[j & 1] + krow * G::BN * 4 + (j >> 1) * 256);
  };

  // One stage out of slot SL. Fragments of step q+1 are requested in front of the MFMAs of step q (the first step of the NEXT stage in front
  // of this stage's last step: no barrier sits between two stages). The stage's ONE barrier is in its middle, in the shadow of an MFMA:
  // behind it stage t+1 has landed for every wave and every wave is past stage t-1, whose slot the requests for stage t+2 may now fill.
  auto stage = [&](auto slc) __attribute__((always_inline)) {
    constexpr int SL = decltype(slc)::value, NS = (SL + 1) % S, FILL = (SL + S - 1) % S;
#pragma unroll
    for (int q = 0; q < NSTEP; ++q) {
      const int m = q >> 2, s = q & 3;
      if (q + 1 < NSTEP) {
        const int m1 = (q + 1) >> 2, s1 = (q + 1) & 3;
        if (s1 == 0) load_a(m1 & 1, ab[SL][m1]);
        load_b((q + 1) & 1, bb_[SL], 8 * m1 + s1);
      } else {
        load_a(0, ab[NS][0]);
        load_b(0, bb_[NS], 0);
      }
      if (q == NSTEP / 2) {
        wait_vm<(S - 3) * NDMA>();  // stage t+1 has landed; t+2 .. t+S-2 may still be in flight
        __builtin_amdgcn_s_barrier();
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // the requests for stage t+2, one piece in front of each MFMA from the barrier on (a burst of them is ~5 scalar instructions per
          // piece during which this wave issues no MFMA)
          const int g = (q - NSTEP / 2) * TM * TN + i * TN + j;
          if (g >= 0 && g < NDMA) issue_piece(FILL, g);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m & 1][i][s], b[q & 1][j], acc[i][j], 0, 0, 0);
        }
    }
  };

#pragma unroll
  for (int sl = 0; sl < S - 1; ++sl) issue(sl);
  wait_vm<(S - 2) * NDMA>();  // stage 0 has landed (stage 1 is waited for at the first mid-stage barrier)
  __builtin_amdgcn_s_barrier();
  load_a(0, ab[0][0]);
  load_b(0, bb_[0], 0);
  int t = 0;
  for (; t + S <= nt; t += S) static_for<0, S>([&](auto i) __attribute__((always_inline)) { stage(i); });
  const int rest = nt - t;  // < S
  static_for<0, S - 1>([&](auto i) __attribute__((always_inline)) {
    if (decltype(i)::value < rest) stage(i);
  });
  wait_vm<0>();  // the re-read requests of the tail must not outlive the workgroup's LDS

  // ---- C: result register r of a 32x32 tile is row (r & 3) + 8 (r >> 2) + 4 kh, column l31: 128-byte row segments per half wave
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float* dst = C + (size_t)row * N + n0 + (wn * TN + j) * 32 + l31;
        if constexpr (KSPLIT) unsafeAtomicAdd(dst, acc[i][j][r]);  // global_atomic_add_f32, no return
        else *dst = acc[i][j][r];
      }
}

__global__ __launch_bounds__(256) void zero_f4_kernel(f4v* __restrict__ p, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) p[i] = f4v{0.f, 0.f, 0.f, 0.f};
}

template <int WM, int WN, int TM, int TN, int BK, int S, bool KSPLIT = false>
int launch(const void* a, const void* b, void* c, int M, int N, int K, int swizzle, hipStream_t st) {
  using G = Geo<WM, WN, TM, TN, BK, S>;
  if (M % G::BM || N % G::BN || K % BK) return CLN_ERR_UNSUPPORTED;
  if (KSPLIT && K < 2 * BK) return CLN_ERR_UNSUPPORTED;
  // the DMA's per-lane source offset is an unsigned 32-bit byte count inside the tile's A rows / the stage's B rows
  if ((unsigned long long)G::BM * (unsigned long long)K * 4ull > 0xFFFFFFFFull || (unsigned long long)BK * (unsigned long long)N * 4ull > 0xFFFFFFFFull)
    return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr attr;
  auto kfn = sgemm_dma_kernel<WM, WN, TM, TN, BK, S, KSPLIT>;
  if (cln_ensure_lds(attr, reinterpret_cast<const void*>(kfn), G::LDS) != CLN_OK) return CLN_ERR_LAUNCH;
  const int tiles_n = N / G::BN, grid = (M / G::BM) * tiles_n;
  if (KSPLIT) {
    // C zeroed by a kernel of ours, not hipMemsetAsync: replayed from a captured graph the runtime's memset node left stale patterns in a quarter
    // of a 4 MB buffer from the second replay on (ROCm 7.2, profiles/r06_sgemm_ksplit.log); M * N is a multiple of 64 * 128 here
    const long long n4 = (long long)M * N / 4;
    CLN_LAUNCH(zero_f4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (f4v*)c, n4);
    if (cln_check_launch() != CLN_OK) return CLN_ERR_LAUNCH;  // (the next CLN_LAUNCH clears the error slot)
  }
  CLN_LAUNCH(kfn, dim3(grid, KSPLIT ? 2 : 1), dim3(G::THREADS), G::LDS, st, (const float*)a, (const float*)b, (float*)c, M, N, K, tiles_n,
             swizzle ? 1 : 0);
  return cln_check_launch();
}

}  // namespace sgemm_dma
