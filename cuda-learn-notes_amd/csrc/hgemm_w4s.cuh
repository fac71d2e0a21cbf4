// One-wave-per-SIMD HGEMM with a `stages`-deep LDS ring of 32-deep K slots (round 4).
//
// Boundary: the `stages` argument of the reference's top NN / TN names (kernels/hgemm/mma/basic/hgemm_mma_stage.cu:2380-2454:
// one kernel template, `case 2/3/4/5` = K_STAGE ring slots of a BK = 32 tile; hgemm.py:359-361 prints stages 2 / 3 / 4 rows).
// Rounds 1-3 answered stages 3 / 4 / 5 on the 256x256 tile with OTHER kernels (8-wave ring / k-half ping-pong: 0.68-0.84x of
// the stages = 2 kernel); this is the stages = 2 kernel's structure (hgemm_w4.cuh: 4 waves, one per SIMD, 128x128 wave
// tiles, 256 accumulator registers tied to AGPRs, both register sets of A / B fragments, LDS-DMA) with the ring depth as a
// real parameter:
//
//   slot      = one 32-deep K step of the 256x256 tile: A image [256 rows][64 B] + B image (TN: [256][64 B]; NN: [32 k][512 B])
//               = 32 KiB; S slots = 64 / 96 / 128 / 160 KiB of LDS (S = 2 / 3 / 4 / 5; the production stages = 2 kernel holds the
//               same 128 KiB as two 64-deep tiles)
//   body(s)   = the 64 MFMAs of slot s (fragments of slot s are in register set s & 1), MFMA index n:
//     n = 1,3,..,31   read the fragments of slot s + 1 into the other register set
//     n = 36      B   lgkmcnt(0) + vmcnt((S - 2) * 8) + s_barrier: every wave has read all of slot s + 1 (its buffer may be
//                     overwritten) and every wave's pieces of slot s + 2 have landed (S - 2 younger slots stay in flight)
//     n = 38 + 8p     LDS-DMA piece p of slot s + 1 + S into the buffer of slot s + 1; pieces 4..7 fall past n = 63 and are issued
//                     by body(s + 1) at n = 6, 14, 22, 30
//   One barrier per 64 MFMAs (as the stages = 2 kernel: two per 128), prefetch distance S - 1 slots.
//   Past the end of K the pieces re-fetch the last slot into dead buffers (uniform counts, no tail code); the ring index is a
//   compile-time constant (the loop is unrolled over lcm(2, S) bodies; K / 32 is even, the loop exits between pairs).
// Every accumulator sees the 32-deep k-steps in ascending order, as in every other kernel of this file set: the result is
// bit-identical to stages = 2 (tests/test_gpu_hgemm.py).
#pragma once
#include "hgemm_w4.cuh"

namespace hgemm {

template <int BM_, int BN_, int LAYOUT_, int S_>
struct W4SCfg {
  static constexpr int BM = BM_, BN = BN_, BK = 32, NW = 4, LAYOUT = LAYOUT_, S = S_;
  static constexpr int WTM = BM / 2, WTN = BN / 2, FM = WTM / 16, FN = WTN / 16;
  static constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, SLOT_BYTES = A_BYTES + B_BYTES, RING_BYTES = S * SLOT_BYTES;
  static constexpr int EPI_BYTES = 4 * 64 * (FN * 32 + 16);
  static constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
  static constexpr int AP = BM / 64, BP = BN / 64, NP = AP + BP;  // 1-KiB DMA pieces per wave per slot
  static constexpr int NR = FM + FN, NM = FM * FN;                // fragment reads / MFMAs per slot
  static_assert(BM == 256 && BN == 256, "256x256 tile");
  static_assert(S >= 2 && S <= 5 && LDS_BYTES <= 160 * 1024, "ring depth");
};

template <int LAYOUT, int S, int EPI = 3>
__global__ __launch_bounds__(256, 1) void hgemm_w4s_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                                           half_t* __restrict__ Cmat, int M, int N, int K, int tiles_m,
                                                           int tiles_n, int swizzle, int band) {
  using C = W4SCfg<256, 256, LAYOUT, S>;
  constexpr int FM = C::FM, FN = C::FN, NR = C::NR, NP = C::NP, NM = C::NM, AP = C::AP;
  constexpr int U = (S % 2 == 0) ? S : 2 * S;  // bodies per unrolled round: register-set parity x ring position
  constexpr int R_FIRST = 1, R_STEP = 2, BPOS = 36, D_FIRST = 38, D_STEP = 8;
  constexpr int D_EARLY = (NM - 1 - D_FIRST) / D_STEP + 1;  // pieces issued inside the body that owns them (4)
  static_assert(R_FIRST + (NR - 1) * R_STEP < BPOS && BPOS < D_FIRST && D_EARLY < NP && D_FIRST + (NP - 1) * D_STEP - NM < BPOS,
                "reads drained before the barrier, DMA after it, late pieces before the next barrier");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  if (swizzle & 4) {  // bit 2: operands larger than the Infinity Cache -- XCDs take the band walk in interleaved chunks (hgemm_mfma.cuh)
    tile_coords_interleaved(blockIdx.x, gridDim.x, tiles_m, tiles_n, band, tm, tn);
  } else {
    tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, swizzle & 1, band, tm, tn);  // bit 1 of `swizzle`: non-temporal C stores allowed (w4_nt_ok)
  }
  const int m0 = tm * C::BM, n0 = tn * C::BN;

  KFill<C, AP> fa;
  fa.init(K, wave, lane);
  KFill<C, C::BP> fbt;
  NFillW<C, C::BP> fbn;
  if constexpr (LAYOUT == TN) fbt.init(K, wave, lane);
  else fbn.init(N, wave, lane);
  const char* a_base = reinterpret_cast<const char*>(A + (size_t)m0 * K);
  const char* b_base = (LAYOUT == TN) ? reinterpret_cast<const char*>(B + (size_t)n0 * K) : reinterpret_cast<const char*>(B + n0);
  const unsigned b_step = (LAYOUT == TN) ? 64u : 64u * (unsigned)N;  // bytes per 32-deep slot (A: 64)
  const int NS = K / 32;
  const unsigned lds0 = lds_addr_of(smem);
  auto voff_of = [&](int p) -> unsigned {
    return p < AP ? fa.voff[p < AP ? p : 0] : (LAYOUT == TN ? fbt.voff[(p - AP) < C::BP ? (p - AP) : 0] : fbn.voff[(p - AP) < C::BP ? (p - AP) : 0]);
  };
  // sources of the slot whose EARLY pieces are issued next (a_cur / b_cur) and of the slot whose LATE pieces are still to be issued (a_old / b_old);
  // past the last slot the sources stay on it (its re-fetch goes to a dead buffer)
  int q_issue = 0;
  const char *a_cur = a_base, *b_cur = b_base, *a_old = a_base, *b_old = b_base;
  auto advance = [&]() {
    a_old = a_cur;
    b_old = b_cur;
    q_issue = q_issue + 1 < NS ? q_issue + 1 : NS - 1;
    a_cur = a_base + (size_t)q_issue * 64u;
    b_cur = b_base + (size_t)q_issue * b_step;
  };
  auto piece = [&](int p, unsigned img) { glds16_asm(p < AP ? a_cur : b_cur, voff_of(p), img + (unsigned)(p * 4 + wave) * 1024u); };
  // in-loop form: M0 walks the destinations (two instructions per piece); nothing else in the K loop uses M0
  auto piece_m0 = [&](int p, unsigned img, bool late, bool set_m0) {
    const char* src = late ? (p < AP ? a_old : b_old) : (p < AP ? a_cur : b_cur);
    if (set_m0)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x1000"
                   :: "v"(voff_of(p)), "s"(src), "s"(img + (unsigned)(p * 4 + wave) * 1024u) : "memory", "scc", "m0");
    else
      asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x1000" :: "v"(voff_of(p)), "s"(src) : "memory", "scc", "m0");
  };

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  // asm MFMAs are invisible to hipcc's hazard pass: pin the zero-filled tiles into their AGPRs here, then pad (hgemm_w4.cuh)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[i][j]));
  asm volatile("s_nop 7");
  h8 af[2][FM], bf[2][FN];
  // fragment read op r of a slot into register set `set`: r = 0 -> A fragment 0, 1..FN -> B fragments, FN+1.. -> A fragments 1..FM-1
  auto read_op = [&](const char* img, int set, int r) {
    const char* bimg = img + C::A_BYTES;
    if (r == 0 || r > FN) {
      const int i = r == 0 ? 0 : r - FN;
      af[set][i] = read_kfrag<32>(img, wm * C::WTM + i * 16 + (lane & 15), lane, 0);
    } else {
      const int j = r - 1;
      if constexpr (LAYOUT == TN) bf[set][j] = read_kfrag<32>(bimg, wn * C::WTN + j * 16 + (lane & 15), lane, 0);
      else bf[set][j] = read_nfrag_w<C::BN>(bimg, wn * C::WTN + j * 16, lane, 0);
    }
  };
#define W4S_PIN() __builtin_amdgcn_sched_barrier(0)
#define W4S_BARRIER()                \
  do {                               \
    __builtin_amdgcn_s_barrier();    \
    asm volatile("" ::: "memory");   \
  } while (0)

  // body u of a round (slot s = u mod U): MFMAs from register set u & 1; reads slot s + 1 from buffer (u + 1) % S into the other set;
  // late pieces of slot s + S go to buffer u % S, early pieces of slot s + 1 + S to buffer (u + 1) % S
  auto body = [&](auto uc) {
    constexpr int u = decltype(uc)::value, P = u & 1, RB = (u + 1) % S, LB = u % S;
    const char* rimg = smem + RB * C::SLOT_BYTES;
    const unsigned rimg_lds = lds0 + RB * C::SLOT_BYTES, limg_lds = lds0 + LB * C::SLOT_BYTES;
    static_for<NM>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int i = n / FN, lo0 = n % FN, j = (i & 1) ? FN - 1 - lo0 : lo0;  // boustrophedon: one operand changes between MFMAs
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bf[P][j]), "v"(af[P][i]));
      constexpr bool R1 = n >= R_FIRST && (n - R_FIRST) % R_STEP == 0 && (n - R_FIRST) / R_STEP < NR;
      constexpr bool DP = n >= D_FIRST && (n - D_FIRST) % D_STEP == 0 && (n - D_FIRST) / D_STEP < NP;
      constexpr int gl = n + NM;  // position of a late piece on the previous body's clock
      constexpr bool LP = (gl - D_FIRST) % D_STEP == 0 && (gl - D_FIRST) / D_STEP < NP && (gl - D_FIRST) / D_STEP >= D_EARLY;
      if constexpr (R1) read_op(rimg, P ^ 1, (n - R_FIRST) / R_STEP);
      if constexpr (LP) piece_m0((gl - D_FIRST) / D_STEP, limg_lds, true, (gl - D_FIRST) / D_STEP == D_EARLY);
      if constexpr (n == BPOS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vmcnt<(S - 2) * NP>();
        W4S_BARRIER();
      }
      if constexpr (DP) piece_m0((n - D_FIRST) / D_STEP, rimg_lds, false, n == D_FIRST);
      if constexpr (R1 || DP || LP || n == BPOS) W4S_PIN();
    });
    advance();
  };

  // prologue: slots 0 .. S-1 requested; slot 0 landed and read; then the barrier of a body "-1": buffer 0 is free, slot 1 has landed
#pragma unroll
  for (int sl = 0; sl < S; ++sl) {
#pragma unroll
    for (int p = 0; p < NP; ++p) piece(p, lds0 + sl * C::SLOT_BYTES);
    advance();
  }
  wait_vmcnt<(S - 1) * NP>();
  W4S_BARRIER();
#pragma unroll
  for (int r = 0; r < NR; ++r) read_op(smem, 0, r);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wait_vmcnt<(S - 2) * NP>();
  W4S_BARRIER();
#pragma unroll
  for (int p = 0; p < D_EARLY; ++p) piece(p, lds0);  // early pieces of slot S (buffer 0); body 0 issues the late ones
  advance();
  W4S_PIN();

  // a_cur / a_old are one slot ahead of what `body` expects at its start (body issues late pieces of a_old, early ones of a_cur, then advances):
  // after the prologue a_old = slot S (late pieces pending), a_cur = slot S + 1 -- exactly the state at the top of body 0.
  int s = 0;
  do {
    static_for<U / 2>([&](auto hc) {
      constexpr int h = decltype(hc)::value;
      if (s < NS) {  // K / 32 is even: bodies go in pairs (register sets 0, 1)
        body(std::integral_constant<int, 2 * h>{});
        body(std::integral_constant<int, 2 * h + 1>{});
        s += 2;
      }
    });
  } while (s < NS);
#undef W4S_PIN
  wait_vmcnt<0>();  // the re-fetches past the end of K: landed (in dead buffers) before the epilogue reuses the LDS
  W4S_BARRIER();
#undef W4S_BARRIER
  // last MFMA results -> v_accvgpr_read: pad, then re-define every tile after the pad (hgemm_w4.cuh)
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) asm volatile("" : "+a"(acc[i][j]));
  store_wide_tile_via_lds<FM, FN, EPI - 2>(Cmat, N, m0 + wm * C::WTM, n0 + wn * C::WTN, lane, smem + wave * (64 * (FN * 32 + 16)), acc, (swizzle >> 1) & 1);
}

// K the structure covers: whole pairs of 32-deep slots, at least 2 S of them
inline bool w4s_k_ok(int K, int S) { return K % 64 == 0 && K / 32 >= 2 * S; }

template <int LAYOUT, int S, int EPI = 3>
int launch_w4s(const void* a, const void* b, void* c, int M, int N, int K, int swizzle, int swizzle_stride, hipStream_t stream) {
  using C = W4SCfg<256, 256, LAYOUT, S>;
  if (M % C::BM || N % C::BN || !w4s_k_ok(K, S)) return CLN_ERR_UNSUPPORTED;
  const int tiles_m = M / C::BM, tiles_n = N / C::BN;
  const int band = (swizzle && swizzle_stride >= C::BN) ? swizzle_stride / C::BN : tiles_n;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&hgemm_w4s_kernel<LAYOUT, S, EPI>), C::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  CLN_LAUNCH((hgemm_w4s_kernel<LAYOUT, S, EPI>), dim3(w4_grid(M, N, K, swizzle, tiles_m, tiles_n)), dim3(256), C::LDS_BYTES, stream, (const half_t*)a,
             (const half_t*)b, (half_t*)c, M, N, K, tiles_m, tiles_n, w4_swizzle_arg(M, N, K, swizzle, tiles_m * tiles_n), band);
  return cln_check_launch();
}

}  // namespace hgemm
