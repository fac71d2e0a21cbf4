// FlashAttention-2 forward for head dims 64 and 128, ONE WAVE PER SIMD with a hand-placed instruction stream
// (BASELINE config C4 = [4,8,2048,64] and its D = 128 sibling). Reference rung:
// kernels/flash-attn/mma/basic/flash_attn_mma_share_qkv.cu:66 ("shared-QKV": Q lives in registers).
//
// Same wave shape as the register-blocked probe (flash_attn_rb.cuh: 4 waves, a wave owns 64 query rows = two 32-row
// groups, every K / V fragment feeds two MFMAs), built with the method that worked for the HGEMM (hgemm_w4.cuh):
//   * every MFMA is inline asm with its accumulator tied to a register class: S^T tiles in VGPRs (the softmax reads
//     them in place, no v_accvgpr_read), O^T tiles and the Q fragments in AGPRs;
//   * the iteration is ONE static_for over its MFMAs; softmax slices, fragment reads, LDS-DMA pieces are compile-time
//     hooks behind chosen MFMAs, pinned with sched_barrier -- an in-order wave hides ~5 single-issue instructions
//     behind each 32-cycle MFMA, so WHERE they sit decides the speed;
//   * iteration j = [ PV(j) MFMAs || row max + rescale decision of S(j+1) ] then [ QK^T(j+2) MFMAs || exponentials of
//     S(j+1) ] (software pipeline over three tiles, two S sets): the first MFMAs after the barrier consume V fragments
//     of a tile that has been visible for a whole iteration (prefetched before the barrier), the K fragments of tile
//     j+2 are read during the PV phase;
//   * Q is pre-multiplied by log2(e)/sqrt(d) and the S^T accumulators START at -m (the deferred running row max)
//     through the MFMA C operand: P = exp2(acc), no per-element fma (flash_attn_dsplit.cuh OPT_PRE);
//   * K/V tiles of 64 keys in a 4-slot LDS ring, LDS-DMA three tiles ahead, ONE workgroup barrier per tile.
// Hazards the compiler cannot see (asm MFMAs) hold by construction: a softmax slice reads an S tile at least two MFMAs
// after the last MFMA that wrote it; P fragments are written at least one PV step before the MFMA that reads them; the
// (cold) rescale path is fenced with s_nop on both sides.
#pragma once
#include <type_traits>

#include "flash_attn_bigd.cuh"
#include "hgemm_w4.cuh"  // static_for

namespace fa2 {

// Flipped to true where the kernel measures faster than the ping-pong kernel on the GPU. Round 2
// (profiles/r02_fa_w4_probe.log, r02_fa_clock_power.log): D = 128 +0..4 % (1020 vs 1006 TF sustained at [4,8,2048,128], 1120
// vs 1098 at [2,32,4096,128]), D = 64 -3..-5 % -- both kernels sit at the ~1300 W package cap (sclk 1.9-2.06 GHz), so the
// halved LDS traffic buys little; not enough to swap the production path. The full FA GPU suite passes with either.
constexpr bool W4_PRODUCTION_D64 = false;
constexpr bool W4_PRODUCTION_D128 = false;
constexpr int W4_VAR_D64 = 8, W4_VAR_D128 = 0;

template <int D>
struct GeoW4 {
  static constexpr int BC = 64, NW = 4, BR = 256, NT = 256, NSTG = 4;
  static constexpr int KB = BC / 32, NK = D / 16, NDB = D / 32, NST = BC / 16;
  static constexpr int ROW = D * 2, TILE = BC * ROW, STAGE = 2 * TILE, RING = NSTG * STAGE;
  static constexpr int OS = D * 2 + 16, EPI = NW * 64 * OS;
  static constexpr int LDS_BYTES = RING > EPI ? RING : EPI;
  static constexpr int PPW = TILE / 1024 / NW, RPP = 1024 / ROW, CPR = ROW / 16;
  static constexpr int NFV = NST * NDB, NFK = KB * NK, NF = NFV + NFK;  // fragments per iteration (two MFMAs each)
  static constexpr int NP = 2 * NFV, NQ = 2 * NFK, NM = NP + NQ;        // MFMAs per iteration
  static_assert(D == 64 || D == 128, "head dims 64 and 128");
  static __device__ __forceinline__ int swz_k(int row) { return CPR >= 16 ? (row & 15) : ((row >> 1) & 7); }
  static __device__ __forceinline__ int swz_v(int row) { return CPR >= 16 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2); }
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ float w4_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__device__ __forceinline__ float w4_max2(float a, float b) {  // no input canonicalisation (fmaxf adds two v_max x, x)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// schedule knobs (MFMA index within the iteration); VAR selects probe variants
struct FaW4Sched {
  int max0, max_per_gap;  // row-max items (one v_max3 each) start / per gap
  int exp0;               // first gap that carries exponentials
  int pd;                 // fragments in flight ahead of their MFMAs
};
constexpr FaW4Sched fa_w4_sched(int D, int var) {
  (void)D;
  switch (var) {
    case 1: return {2, 8, 7, 4};
    case 2: return {2, 4, 11, 2};
    default: return {2, 4, 11, 4};
  }
}

template <int D, int VAR = 0, int ABL = 0>
__global__ __launch_bounds__(256, 1) void fa2_fwd_w4_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                            const half_t* __restrict__ V, half_t* __restrict__ O,
                                                            int N, int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoW4<D>;
  constexpr FaW4Sched S = fa_w4_sched(D, VAR & 7);
  // VAR & 8: the row sums of P ride on the matrix pipe -- one extra "d block" of V^T made of ones per 16-key step (two
  // MFMAs, no LDS fragment) instead of 64 v_add per lane and iteration: the kernel is VALU-issue-bound, not MFMA-bound
  constexpr bool SUMM = (VAR & 8) != 0;
  constexpr int PD = S.pd, NF = G::NF, NFV = G::NFV, NK = G::NK, NDB = G::NDB;
  constexpr int NDBV = NDB + (SUMM ? 1 : 0);       // PV MFMA pairs per 16-key step
  constexpr int NPVV = G::NST * NDBV;              // "virtual" V fragments per iteration (the ones block has no fragment)
  constexpr int NP = 2 * NPVV, NM = NP + G::NQ;    // MFMAs per iteration
  constexpr int NMAX = 32, NPAIR = 32;  // v_max3 items / exponential pairs per iteration (64 scores per lane)
  constexpr int DECIDE_AT = S.max0 + (NMAX + S.max_per_gap - 1) / S.max_per_gap;  // gap of the rescale decision
  static_assert(S.exp0 > DECIDE_AT && S.exp0 < NM && DECIDE_AT < NP, "exponentials need the decided running max; the decision falls in the PV phase");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  int head_i, qbi;
  {
    const int bid = blockIdx.x;
    if ((n_heads & 7) == 0) {  // heads pinned to XCDs: a head's K/V stays in one L2
      const int xcd = bid & 7, k = bid >> 3;
      head_i = (k / n_qblk) * 8 + xcd;
      qbi = k - (k / n_qblk) * n_qblk;
    } else {
      head_i = bid / n_qblk;
      qbi = bid - head_i * n_qblk;
    }
  }
  const size_t head = (size_t)head_i * N * D;
  const int q_row0 = qbi * G::BR + wave * 64;
  const unsigned lds0 = hgemm::lds_addr_of(smem);

  // ---- LDS-DMA (same lane-linear images and source-side swizzles as flash_attn_dsplit.cuh / flash_attn_rb.cuh)
  const char* src_k = reinterpret_cast<const char*>(K + head);
  const char* src_v = reinterpret_cast<const char*>(V + head);
  const int lr = lane / G::CPR, lc = lane % G::CPR, rlow = wave * G::RPP + lr;
  const unsigned src_lane_k = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ G::swz_k(rlow)) << 4);
  const unsigned src_lane_v = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ G::swz_v(rlow)) << 4);
  // IN_LOOP: M0 is set and left (nothing else in the KV loop uses M0) -- three instructions instead of six
  auto dma_piece = [&](int n, int jt, int slot, bool in_loop = false) __attribute__((always_inline)) {  // n < PPW: K piece n; else V piece n - PPW
    if constexpr ((ABL & 1) != 0) return;
    const int op = n >= G::PPW, i = op ? n - G::PPW : n;
    const int piece = i * 4 + wave;
    const unsigned voff = op ? src_lane_v : (src_lane_k ^ (unsigned)(((i * 4 * G::RPP) & 15) << 4));
    const char* s = (op ? src_v : src_k) + (size_t)jt * G::TILE + piece * 1024;
    const unsigned dst = lds0 + slot * G::STAGE + op * G::TILE + piece * 1024;
    if (in_loop) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(s), "s"(dst) : "memory");
    else hgemm::glds16_asm(s, voff, dst);
  };

  const int T = N / G::BC;  // N % 256 == 0 (launcher): T is a multiple of 4
  // ---- Q fragments (B operand of S^T = K Q^T), pre-scaled, parked in the accumulator half of the register file
  h8 qf[2][NK];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const half_t* qp = Q + head + (size_t)(q_row0 + g * 32 + l31) * D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) qf[g][ks] = *reinterpret_cast<const h8*>(qp + ks * 16);
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int n = 0; n < 2 * G::PPW; ++n) dma_piece(n, t, t);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible: Q and tiles 0..2 are in
  {
    const half_t sc = (half_t)scale_log2e;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) qf[g][ks] = qf[g][ks] * sc;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) asm volatile("" : "+a"(qf[g][ks]));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // fragment offsets (lane constants; see flash_attn_dsplit.cuh for the swizzle algebra)
  const int kbase = l31 * G::ROW + ((hi ^ G::swz_k(l31)) << 4);
  const int i16 = lane & 15;
  const int v_row = 4 * hi + (i16 >> 2);
  const int vbase = v_row * G::ROW + ((G::swz_v(v_row) + (((lane >> 4) & 1) * 2) + ((i16 & 3) >> 1)) << 4) + ((i16 & 1) << 3);
  // K fragment fk = ks * 2 + kb (k-step major: consecutive MFMAs on the same S accumulator are four apart -- two apart,
  // a dependent MFMA waits for its predecessor): keys kb*32 + l31, k-step ks
  auto k_frag = [&](int kb_j, int fk) __attribute__((always_inline)) {
    const int kb = fk % 2, ks = fk / 2;
    return *reinterpret_cast<const h8*>(smem + (kb_j ^ ((ks & 7) << 5)) + (ks >> 3) * 256 + kb * 32 * G::ROW);
  };
  auto v_frag = [&](int vb_j, int fv) __attribute__((always_inline)) {  // fragment fv = st * NDB + b: keys 16*st + v_row and + 8, d block b
    const int st = fv / NDB, b = fv % NDB;
    const char* vp = smem + (vb_j ^ ((b & 3) << 6)) + (16 * st) * G::ROW + (b >> 2) * 256;
    return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::ROW));
  };

  f16v ot[2][NDB];  // O^T accumulators: AGPRs
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[g][b][r] = 0.f;
      asm volatile("" : "+a"(ot[g][b]));
    }
  f16v lt[2];  // SUMM: row sums of P, accumulated by the matrix pipe (all 16 registers of a lane carry its row's sum)
  h8 ones;
  if constexpr (SUMM) {
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (half_t)1.0f;
    asm volatile("" : "+v"(ones));
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int r = 0; r < 16; ++r) lt[g][r] = 0.f;
      asm volatile("" : "+a"(lt[g]));
    }
  }
  f16v s_acc[2][2][2];  // [set][row group][key block]: VGPRs
  f16v minit[2];        // -m in all 16 registers of a lane (the chains of a tile start here)
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[g][r] = 0.f;
    asm volatile("" : "+v"(minit[g]));
  }
  asm volatile("s_nop 7");  // zero-fill (VALU / accvgpr writes) -> first asm MFMA
  h8 pf[2][G::NST];
  h8 fr[PD];  // fragment ring: V fragments of tile j, then K fragments of tile j+2, ...
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f}, mx[2] = {0.f, 0.f};
  float mxp[2][4];  // partial row maxima: four independent v_max3 chains per row group (a single chain is latency-bound)

  // ---- the pieces of work
  auto qk_mfma = [&](auto set_c, int fk, int g, const h8& kf) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
    const int kb = fk % 2, ks = fk / 2;
    auto& sa = s_acc;  // (asm operands alone do not make a generic lambda capture its locals)
    auto& qa = qf;
    auto& mi = minit;
    if (ks == 0)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(sa[set][g][kb]) : "v"(kf), "a"(qa[g][0]), "v"(mi[g]));
    else
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sa[set][g][kb]) : "v"(kf), "a"(qa[g][ks]));
  };
  auto pv_mfma = [&](int st, int b, int g, const h8& vf) __attribute__((always_inline)) {
    if (b < NDB) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ot[g][b < NDB ? b : 0]) : "v"(vf), "v"(pf[g][st]));
    else if constexpr (SUMM) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(lt[g]) : "v"(ones), "v"(pf[g][st]));
  };
  // row-max item i of score set `set`: i = g*16 + kb*8 + q covers s[g][kb][2q], [2q+1]
  auto max_item = [&](auto set_c, int i) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
    const int g = i >> 4, kb = (i >> 3) & 1, q = i & 7;
    const float x0 = s_acc[set][g][kb][2 * q], x1 = s_acc[set][g][kb][2 * q + 1];
    const int c = i & 3, first = (i & 15) < 4;  // chain c of row group g takes items c, c+4, c+8, c+12
    if (first) mxp[g][c] = w4_max2(x0, x1);
    else mxp[g][c] = w4_max3(mxp[g][c], x0, x1);
    if ((i & 15) == 15) mx[g] = w4_max2(w4_max2(mxp[g][0], mxp[g][1]), w4_max2(mxp[g][2], mxp[g][3]));
  };
  // exponential pair p of score set `set` (P-step major: p = st*8 + g*4 + q): scores s[g][st>>1][(st&1)*8 + 2q], +1
  float psum[2] = {0.f, 0.f};
  float alpha_pend[2] = {1.f, 1.f};
  bool pend = false;  // wave-uniform: a rescale was decided, O^T / l not yet scaled
  auto exp_pair = [&](auto set_c, int p) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
    const int st = p >> 3, g = (p >> 2) & 1, q = p & 3, r = (st & 1) * 8 + 2 * q;
    const float x0 = s_acc[set][g][st >> 1][r], x1 = s_acc[set][g][st >> 1][r + 1];
    const float a0 = (ABL & 2) ? x0 : __builtin_amdgcn_exp2f(x0);
    const float a1 = (ABL & 2) ? x1 : __builtin_amdgcn_exp2f(x1);
    if constexpr (!SUMM) psum[g] += a0 + a1;
    const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
    pf[g][st][2 * q] = a[0], pf[g][st][2 * q + 1] = a[1];
  };
  // running max / rescale once a tile's row maxima are known (the scores are relative to the current -m)
  auto decide = [&](auto set_c, auto first_tag) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    float d[2];
    bool grow = false;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx[g]), __float_as_uint(mx[g]), false, false);
      d[g] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      grow |= d[g] > 8.0f;  // deferred: rescale only when a row grew by more than 2^8
    }
    if (FIRST || __builtin_expect(__builtin_amdgcn_ballot_w64(grow) != 0, 0)) {
      asm volatile("s_nop 15\n\ts_nop 7\n\t; rescale" ::: "memory");  // cold; in-flight PV MFMAs retire before O^T is read
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float delta = FIRST ? d[g] : fmaxf(d[g], 0.f);
        m_run[g] += delta;  // tile 0: m_run was 0
        // O^T and the row sums are scaled LATER (apply_rescale, behind the last PV MFMA of this iteration): the PV MFMAs
        // still to come add P(j), which is relative to the OLD max -- scaling now would leave them unscaled
        if constexpr (!FIRST) alpha_pend[g] = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s_acc[set][g][kb][r] -= delta;  // accumulated from the old -m
#pragma unroll
        for (int r = 0; r < 16; ++r) minit[g][r] = -m_run[g];
        auto& sa = s_acc;
        auto& mi = minit;
        asm volatile("" : "+v"(mi[g]), "+v"(sa[set][g][0]), "+v"(sa[set][g][1]));
      }
      asm volatile("s_nop 7" ::: "memory");  // VALU writes -> asm MFMA reads
      if constexpr (!FIRST) pend = true;
    }
  };
  auto apply_rescale = [&]() __attribute__((always_inline)) {
    if (__builtin_expect(pend, 0)) {
      asm volatile("s_nop 15\n\ts_nop 7\n\t; apply rescale" ::: "memory");  // the PV MFMAs of this iteration retire first
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float alpha = alpha_pend[g];
        l_run[g] *= alpha;
        if constexpr (SUMM) {
#pragma unroll
          for (int r = 0; r < 16; ++r) lt[g][r] *= alpha;
          asm volatile("" : "+a"(lt[g]));
        }
#pragma unroll
        for (int b = 0; b < NDB; ++b) {
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[g][b][r] *= alpha;
          asm volatile("" : "+a"(ot[g][b]));
        }
      }
      asm volatile("s_nop 7" ::: "memory");
      pend = false;
    }
  };

#define FW4_PIN() __builtin_amdgcn_sched_barrier(0)
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // ---- prologue: S(0) -> max -> first running max; S(1); P(0); the first V fragments of tile 0
  {
#pragma unroll
    for (int fk = 0; fk < G::NFK; ++fk) {
      const h8 kf = k_frag(kbase, fk);
      qk_mfma(I0{}, fk, 0, kf);
      qk_mfma(I0{}, fk, 1, kf);
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s_acc[0][0][0]), "+v"(s_acc[0][0][1]), "+v"(s_acc[0][1][0]), "+v"(s_acc[0][1][1]));
#pragma unroll
    for (int i = 0; i < NMAX; ++i) max_item(I0{}, i);
    decide(I0{}, std::true_type{});
#pragma unroll
    for (int fk = 0; fk < G::NFK; ++fk) {
      const h8 kf = k_frag(kbase + G::STAGE, fk);
      qk_mfma(I1{}, fk, 0, kf);
      qk_mfma(I1{}, fk, 1, kf);
    }
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) exp_pair(I0{}, p);
    l_run[0] += psum[0], l_run[1] += psum[1];
    psum[0] = psum[1] = 0.f;
#pragma unroll
    for (int i = 0; i < PD; ++i) fr[i] = v_frag(vbase + G::TILE, i);
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s_acc[1][0][0]), "+v"(s_acc[1][0][1]), "+v"(s_acc[1][1][0]), "+v"(s_acc[1][1][1]));
    FW4_PIN();
  }

  unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ABL & 32: s_memtime at phase boundaries of iteration 16
  auto mark = [&](int j, int i) __attribute__((always_inline)) {
    if constexpr ((ABL & 32) != 0)
      if (j == 16) stamp[i] = __builtin_amdgcn_s_memtime();
  };
  int slot_j = 0;  // ring slot of tile j
  // One iteration j. CUR = set of S(j+1) (max + exponentials here), the other set receives S(j+2).
  // N1: tile j+1 exists; N2: tile j+2 exists.
  auto iteration = [&](int j, auto cur_c, auto n1_c, auto n2_c) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr bool N1 = decltype(n1_c)::value, N2 = decltype(n2_c)::value;
    using CurT = std::integral_constant<int, CUR>;
    using NxtT = std::integral_constant<int, 1 - CUR>;
    const int s1 = (slot_j + 1) & 3, s2 = (slot_j + 2) & 3, s3 = (slot_j + 3) & 3;
    const int jd = j + 3 < T ? j + 3 : T - 1;  // past the end: refill a dead slot with the last tile (branch-free)
    const int vb_j = vbase + slot_j * G::STAGE + G::TILE;   // V of tile j
    const int vb_n = vbase + s1 * G::STAGE + G::TILE;       // V of tile j+1 (next iteration's first fragments)
    const int kb_j = kbase + s2 * G::STAGE;                 // K of tile j+2
    // own DMA pieces of tile j+2 (issued one iteration ago) landed; behind the barrier everyone's are visible and
    // everyone is done with the slot of tile j-1 (refilled below)
    mark(j, 0);
    hgemm::wait_vmcnt<0>();
    mark(j, 1);
    if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    mark(j, 2);
    hgemm::static_for<NM>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      if constexpr (n == NP / 2) mark(j, 3);
      if constexpr (n == NP) mark(j, 4);
      if constexpr (n == NP && N1) apply_rescale();  // before the first QK^T MFMA: every PV MFMA of the iteration is issued
      if constexpr (n == NP + G::NQ / 2) mark(j, 5);
      constexpr int qv = n / 2, g = n & 1;  // virtual fragment index within the iteration, row group
      constexpr int st = qv / NDBV, b = qv % NDBV;         // PV phase: 16-key step, d block (b == NDB: the ones block)
      constexpr bool REAL = n >= NP || b < NDB;            // consumes a fragment from the ring
      constexpr int q = n < NP ? st * NDB + (b < NDB ? b : NDB - 1) : NFV + (qv - NPVV);  // real fragment index
      if constexpr (n < NP) {
        pv_mfma(st, b, g, fr[q % PD]);
      } else if constexpr (N2) {
        qk_mfma(NxtT{}, q - NFV, g, fr[q % PD]);
      }
      bool hook = false;
      // fragment q + PD into the ring slot just freed (after the second MFMA of fragment q)
      if constexpr (g == 1 && REAL) {
        constexpr int qn = q + PD;
        if constexpr ((ABL & 4) != 0) {  // ablation: no fragment reads
        } else if constexpr (qn < NFV) fr[q % PD] = v_frag(vb_j, qn), hook = true;
        else if constexpr (qn < NF) { if constexpr (N2) fr[q % PD] = k_frag(kb_j, qn - NFV), hook = true; }
        else if constexpr (N1) fr[q % PD] = v_frag(vb_n, qn - NF), hook = true;
      }
      if constexpr (N1) {
        // row max of S(j+1)
        if constexpr (n >= S.max0 && n < DECIDE_AT && !(ABL & 16)) {
#pragma unroll
          for (int i = (n - S.max0) * S.max_per_gap; i < (n - S.max0 + 1) * S.max_per_gap && i < NMAX; ++i) max_item(CurT{}, i);
          hook = true;
        }
        if constexpr (n == DECIDE_AT && !(ABL & 16)) decide(CurT{}, std::false_type{}), hook = true;
        // exponentials of S(j+1), spread evenly over the gaps exp0 .. NM-1
        if constexpr (n >= S.exp0) {
          constexpr int G0 = n - S.exp0, NG = NM - S.exp0;
#pragma unroll
          for (int p = G0 * NPAIR / NG; p < (G0 + 1) * NPAIR / NG; ++p) exp_pair(CurT{}, p);
          hook = true;
        }
      }
      // LDS-DMA of tile j+3: 2*PPW pieces, evenly over the iteration
      if constexpr (n % (NM / (2 * G::PPW)) == 1 && n / (NM / (2 * G::PPW)) < 2 * G::PPW) {
        dma_piece(n / (NM / (2 * G::PPW)), jd, s3, true);
        hook = true;
      }
      (void)hook;
      FW4_PIN();
    });
    mark(j, 6);
    if constexpr (N1) {
      l_run[0] += psum[0], l_run[1] += psum[1];
      psum[0] = psum[1] = 0.f;
    }
    slot_j = s1;
  };
  constexpr std::true_type Y{};
  constexpr std::false_type NO{};
  // tiles 0 .. T-3 in pairs (T % 4 == 0), then the two tail iterations
  for (int j = 0; j + 2 < T; j += 2) {
    iteration(j, I1{}, Y, Y);      // S(j+1) in set 1 (j even), S(j+2) -> set 0
    iteration(j + 1, I0{}, Y, Y);
  }
  iteration(T - 2, I1{}, Y, NO);
  iteration(T - 1, I0{}, NO, NO);
#undef FW4_PIN

  // ---- epilogue: O = O^T / l, staged through LDS (wave-private rows), 16-byte row segments out
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int b = 0; b < NDB; ++b) asm volatile("" : "+a"(ot[g][b]));
  if constexpr (SUMM) asm volatile("" : "+a"(lt[0]), "+a"(lt[1]));
  __builtin_amdgcn_s_barrier();  // every wave is past its last fragment read: the ring becomes the staging area
  asm volatile("" ::: "memory");
  const int lane_e = cln_fresh_lane(), l31_e = lane_e & 31, hi_e = lane_e >> 5;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    float l_tot;
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[g]), __float_as_uint(l_run[g]), false, false);
      l_tot = SUMM ? lt[g][0] : __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l_tot;
    char* ob = smem + (wave * 2 + g) * (32 * G::OS);
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[g][b][rq * 4 + e] * inv);
        *reinterpret_cast<h4*>(ob + l31_e * G::OS + (b * 32 + rq * 8 + hi_e * 4) * 2) = o;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int LPR = D / 8;  // 16-byte segments per row
    half_t* og = O + head + (size_t)(q_row0 + g * 32) * D;
#pragma unroll 4
    for (int it = 0; it < (32 * LPR) / 64; ++it) {
      const int idx = it * 64 + lane_e;
      const int row = idx / LPR, c = idx % LPR;
      *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
    }
  }
  if constexpr ((ABL & 32) != 0) {
    if (blockIdx.x == 0 && lane == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(O);
#pragma unroll
      for (int i = 0; i < 8; ++i) dbg[wave * 8 + i] = stamp[i];
    }
  }
}

// (probe builds with ABL & 32 overwrite the head of O with the time stamps of workgroup 0)
template <int D, int VAR = 0, int ABL = 0>
int launch_fa_w4(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoW4<D>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_w4_kernel<D, VAR, ABL>), G::LDS_BYTES) != CLN_OK)
    return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_w4_kernel<D, VAR, ABL>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
