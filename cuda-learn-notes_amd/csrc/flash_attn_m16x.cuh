// FlashAttention-2 forward, head dims 64 / 128: the 16x16x32 ping-pong kernel of flash_attn_m16.cuh with a SUM-CHECKED
// OPTIMISTIC softmax (round 3). Reference rung: kernels/flash-attn/mma/basic/flash_attn_mma_share_qkv.cu:66.
//
// What changes against flash_attn_m16.cuh, and why. That kernel computes, per KV tile, the row maximum of all scores
// (a serial v_max3 chain + two cross-lane swaps, ~40 of its ~190 VALU instructions per wave and tile), decides whether
// the running reference m must move, and only then starts the exponentials: inside a wave the QK^T MFMAs, the maximum
// and the exponentials are strictly serial. But the maximum is only ever USED to keep P = 2^(s - m) inside fp16
// (the reference m is deferred anyway: it moves only when a row grew by more than 2^8). So:
//   * the S^T accumulators start at -m (as before) and every 16-key block is exponentiated AS SOON AS its MFMA chain
//     has finished, relative to the current reference, while the MFMAs of the next key block run -- no maximum first;
//   * the row sums that the softmax needs anyway are the overflow check: if the per-lane partial sum of a tile is
//     <= 2^15, every P of it is <= 2^15 and fits fp16; inf / NaN fail the comparison too. One compare per query block
//     and tile replaces the max chain;
//   * only when a lane fails the check (or on tile 0, which has no reference yet) the wave takes the COLD path: the true
//     row maxima of the tile, the standard rescale of O / l / the pending scores, and the exponentials again.
//   * the exponentials of the last NDEF key blocks are issued in phase B under the PV MFMAs (their scores are checked
//     directly against 14 at the end of phase A: 4 v_max3 per block instead of a sum that does not exist yet).
// fp16 has the same relative precision from 2^-14 to 2^15 and O / l accumulate in fp32, so a reference that lags the true
// maximum by up to 15 binades costs nothing (the deferred form already allowed 8); the reference never moves down, and
// tile 0 adopts its true maximum, so the largest P of a row is always >= 1.
// Phase A (QK^T + exponentials) is now VALU-dense and phase B (PV) MFMA-only: with OX & 1 the wave raises its priority
// in phase A so its VALU instructions win the issue arbitration against the partner's MFMAs (the matrix pipe needs one
// issue slot per 16 cycles).
#pragma once
#include "flash_attn_m16.cuh"

namespace fa2 {

// OX bits. Production uses M16X_PRIO | M16X_SPLIT_PROLOGUE (= 5); everything from M16X_PRIO_STATIC up is instantiated only in the
// probe library (probe/flash_attn_m16x_probe.hip) and its measured effect is in profiles/r03_fa_c4_ablation_probe.log.
enum : int {
  M16X_PRIO = 1,            // s_setprio 1 in phase A (VALU-dense), 0 in phase B
  M16X_PRIO_B = 2,          // the opposite flip
  M16X_SPLIT_PROLOGUE = 4,  // tile 0's DMA pieces before the Q loads; group 1 does not hold up the first barrier
  M16X_PRIO_STATIC = 8,     // s_setprio 1 once for the group that runs a phase behind, no flips
  M16X_NT_STORE = 16,       // non-temporal O stores
  // ablations (results are garbage by design): no K fragment reads, no V fragment reads, no exponentials, no LDS-DMA after the
  // prologue, no workgroup barriers inside the KV loop (the last only together with the LDS ablations)
  M16X_ABL_K = 32, M16X_ABL_V = 64, M16X_ABL_EXP = 128, M16X_ABL_DMA = 256, M16X_ABL_BAR = 1024,
  // QK^T steps of TWO key blocks interleaved (kb, kb+1 at k-step 0, then both at k-step 1): a dependent MFMA on one accumulator
  // then sits 2 * NQB MFMAs behind the one it depends on instead of NQB (D = 64: NKS = 2)
  M16X_PAIRED_QK = 512,
  M16X_SNAKE = 2048,  // query blocks in snake order (0,1 | 1,0 | ...): every MFMA shares one operand register set with its predecessor
  M16X_FINE = 4096,   // one softmax item behind EACH MFMA (M V M V) instead of the step's MFMAs first and its items after them
  // phase A exponentiates its NOPT blocks behind the LAST NOPT blocks' MFMAs instead of the first ones: a wave's softmax-carrying stretch is
  // then [second half of A, first half of B] and its bare-MFMA stretch [second half of B, first half of A]; the partner group runs one phase
  // behind, so one wave of a SIMD is always in its bare stretch while the other carries softmax work. With M16X_PRIO the priority follows
  // the stretches (1 while carrying softmax work) instead of the phases.
  M16X_LATE = 8192,
  // the deferred key blocks are not checked by their raw scores at the end of phase A (NDEF * NQB * 4 scores through v_max3 per lane and
  // tile); instead their OWN partial row sums are compared once, in phase B, right after their last exponential and before the first PV
  // MFMA that consumes them -- a failing wave rescales O / l there (the PV products of the optimistic blocks are already in O, relative
  // to the old reference, and are scaled with it) and exponentiates the deferred blocks again
  M16X_LATE_CHECK = 16384,
  // the `stages = 1` form (reference kStage = 1, flash_attn_mma_share_qkv.cu:711-762: a tile is requested, waited for, then used): the
  // SAME kernel, same LDS image and arithmetic (bit-identical output), but a wave issues all its pieces of tile j + 1 in ONE burst and
  // waits for them right there (s_waitcnt vmcnt(0)): no request of the wave is in flight while it computes. M16X_ONE_POS (probe) picks the
  // burst's place in the iteration: 0 = top of phase A, 1 = end of phase A, 2 = top of phase B, 3 = end of phase B.
  M16X_ONE_STAGE = 32768,
  M16X_ONE_POS_SHIFT = 16,  // two bits
  // scores scaled in fp32 (round 4; the reference's *_acc_f32 names): Q goes into the MFMAs as loaded, S^T accumulates from 0 and every score takes
  // one v_fma_f32 (s * log2(e)/sqrt(d) - m) on its way into v_exp_f32 -- no fp16 rounding of Q * log2(e)/sqrt(d) (2^-11 relative per term, which
  // amplified keys turn into 3.9e-3 on O where this form has 2e-3), 64 more VALU instructions per wave and 128-key tile at D = 64
  M16X_FSCALE = 1 << 18,
  // probe (round 4; the "one untested trade" of DESIGN_LOG 9.3): ROW SUMS ON THE MATRIX PIPE. l = sum of the fp16 P values, from one more MFMA per
  // 32-key step and query block with an all-ones A operand (every row of its 16x16 result is the row-sum vector, so no cross-lane reduction at the
  // end either); the 64 v_add_f32 per tile go, and the fp16-overflow check of the optimistic blocks becomes a running v_max3 of the exponents (<= 15).
  M16X_MFMA_SUM = 1 << 19,
  // probe (round 5): the partial row sums by v_dot2_f32_f16 -- acc += p0 + p1 over the fp16-ROUNDED pair in ONE instruction instead of two v_add_f32 (64 -> 32
  // row-sum instructions per wave and 128-key tile at D = 64); l becomes the sum of the same fp16 values the numerator uses, an fp16 overflow still fails the check (inf)
  M16X_DOT2_SUM = 1 << 20,
  // probe (round 6, VERDICT r5 #1b "C4's fixed cost"): every wave stamps s_memrealtime (100 MHz, chip-wide) and s_memtime (shader clock) at kernel entry,
  // at the top of the KV loop (prologue done), behind the loop and behind its last O store (waited for), plus its HW_ID / XCC_ID -- 10 words per wave into
  // g_m16x_stamps (probe library: cln_probe_set_stamps). tools/fa_c4_stamps.py turns them into the launch ramp, prologue, per-tile and tail times per CU.
  M16X_STAMP = 1 << 21,
  M16X_ONE_POS = 2           // the shipped position: top of phase B (the MFMA-only phase), 0.95-1.0x of stages = 2 (profiles/r04_fa_one_stage_probe.log)
};

inline unsigned long long* g_m16x_stamps = nullptr;  // device buffer of the M16X_STAMP probe form (10 words per wave); never set in the product library

template <int D_, int RPW_, int BC_, int PD = 4, int NDEF = 1, int OX = 0, bool VT = false>
__global__ __launch_bounds__(512, 2) void fa2_fwd_m16x_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                              const half_t* __restrict__ V, half_t* __restrict__ O,
                                                              int N, int n_qblk, int n_heads, float scale_log2e, unsigned long long* stamps) {
  constexpr bool STAMP = (OX & M16X_STAMP) != 0;
  unsigned long long st_rt[4] = {0, 0, 0, 0}, st_mt[4] = {0, 0, 0, 0};
  auto stamp = [&](int i) __attribute__((always_inline)) {
    if constexpr (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      st_rt[i] = __builtin_amdgcn_s_memrealtime();
      st_mt[i] = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);
  using G = GeoM16<D_, RPW_, BC_>;
  constexpr int D = G::D, NKB = G::NKB, NKS = G::NKS, NQB = G::NQB, NU = G::NU, NDB = G::NDB, NQK = G::NQK, NPV = G::NPV;
  constexpr int NOPT = NKB - NDEF;            // key blocks exponentiated in phase A
  constexpr int NPAIR = NQB * 2;              // (query block, register pair) items of one key block
  constexpr int PER_STEP = (NPAIR + NKS - 1) / NKS;
  constexpr int DSTEPS = (NOPT / 2) * NDB;    // PV steps before the first P^T k-step that contains a deferred block
  constexpr int DRATE = (NDEF * NPAIR + DSTEPS - 1) / DSTEPS;
  static_assert(NDEF >= 1 && NOPT >= 2, "at least one P^T k-step must be complete at the end of phase A");
  static_assert((OX & M16X_FSCALE) == 0 || (OX & M16X_LATE_CHECK) == 0, "the fp32-scaled form has no late-check variant");
  static_assert((OX & M16X_MFMA_SUM) == 0 || (OX & M16X_LATE_CHECK) == 0, "row sums on the matrix pipe: no late-check variant");
  constexpr bool MS = (OX & M16X_MFMA_SUM) != 0;
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // 1024: the fragment addresses XOR bits 4 .. 8 into (LDS address of smem + offset)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g4 = lane >> 4;
  const int grp = wave >> 2, widx = wave & 3;

  int head_i, qb_i;
  {
    const int bid = blockIdx.x;
    if ((n_heads & 7) == 0) {  // heads pinned to XCDs: a head's K/V stays in one L2
      const int xcd = bid & 7, k = bid >> 3;
      head_i = (k / n_qblk) * 8 + xcd;
      qb_i = k - (k / n_qblk) * n_qblk;
    } else {
      head_i = bid / n_qblk;
      qb_i = bid - head_i * n_qblk;
    }
  }
  const size_t head = (size_t)head_i * N * D;
  const int q_row0 = qb_i * G::BR + wave * G::RPW;
  const unsigned lds0 = hgemm::lds_addr_of(smem);

  // VT (the three *_swizzle_qkv names that take V as [B,H,D,N], reference flash_attn_mma_share_qkv.cu swizzle_qkv form): the V image of a
  // tile is D rows (one per d) of BC keys = RV bytes; a row is contiguous in memory, rows are N * 2 bytes apart. Chunk swizzle by row as
  // the K image of the same row length: row & 15 (256-byte rows), (row >> 1) & 7 (128-byte rows).
  constexpr int RV = G::BC * 2, CPRV = RV / 16, RPPV = 1024 / RV;
  auto swz_vt = [](int row) { return RV == 128 ? (row >> 1) & 7 : row & 15; };
  const char* src_h = reinterpret_cast<const char*>((grp == 0 ? K : V) + head);
  const bool vt_loader = VT && grp == 1;
  const int lr = vt_loader ? lane / CPRV : lane / G::CPR, lc = vt_loader ? lane % CPRV : lane % G::CPR;
  const int sw_src = grp == 0 ? G::swz_k(widx * G::RPP + lr) : VT ? swz_vt(widx * RPPV + lr) : G::swz_v(widx * G::RPP + lr);
  const unsigned src_lane = vt_loader ? (unsigned)lr * (unsigned)N * 2u + (unsigned)((lc ^ sw_src) << 4)
                                      : (unsigned)(lr * G::ROW) + (unsigned)((lc ^ sw_src) << 4);
  auto dma_piece = [&](int jt, int slot, int i) __attribute__((always_inline)) {
    const int piece = i * 4 + widx;
    const char* s = vt_loader ? src_h + (size_t)jt * RV + (size_t)(piece * RPPV) * (size_t)N * 2u : src_h + (size_t)jt * G::TILE + piece * 1024;
    hgemm::glds16_asm(s, src_lane, lds0 + slot * G::STAGE + grp * G::TILE + piece * 1024);
  };

  h8 qf[NQB][NKS];
  const int T = N / G::BC;
  __builtin_assume(T > 0);
  if constexpr ((OX & M16X_SPLIT_PROLOGUE) != 0) {
    // tile 0's pieces first (group 0: K, needed by the first MFMA; group 1: V, needed one phase later), then Q
#pragma unroll
    for (int i = 0; i < G::PPW; ++i) dma_piece(0, 0, i);
  }
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const half_t* qp = Q + head + (size_t)(q_row0 + qb * 16 + i16) * D + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = *reinterpret_cast<const h8*>(qp + ks * 32);
  }
  f4 ot[NDB][NQB];
#pragma unroll
  for (int b = 0; b < NDB; ++b)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) ot[b][qb] = f4{0.f, 0.f, 0.f, 0.f};
  float m_run[NQB], l_run[NQB];
  f4 lacc[NQB];  // MS: the row sums, accumulated by the matrix pipe
  h8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (half_t)1.0f;
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) lacc[qb] = f4{0.f, 0.f, 0.f, 0.f};
  f4 minit[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    m_run[qb] = 0.f, l_run[qb] = 0.f;
    minit[qb] = f4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(minit[qb]));
  }
  if constexpr ((OX & M16X_SPLIT_PROLOGUE) == 0) {
#pragma unroll
    for (int i = 0; i < G::PPW; ++i) dma_piece(0, 0, i);
  }
  constexpr bool FS = (OX & M16X_FSCALE) != 0;
  auto scale_q = [&]() __attribute__((always_inline)) {
    if constexpr (FS) return;  // Q as loaded: the scale is applied to the fp32 scores
    const half_t sc = (half_t)scale_log2e;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        qf[qb][ks] = qf[qb][ks] * sc;
        asm volatile("" : "+v"(qf[qb][ks]));
      }
  };
  if constexpr ((OX & M16X_SPLIT_PROLOGUE) != 0) {
    // group 0 needs K tile 0 (its own pieces) and its Q rows; group 1's V pieces and Q rows are not needed before the
    // second barrier, so group 1 does not hold up the first one
    if (grp == 0) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      scale_q();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp == 1) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      scale_q();
    }
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible: also retires the Q loads
    scale_q();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // LDS byte addresses with the symbol's address folded in ONCE (common.h lds_ld: `smem + offset` costs a v_add_u32 of the relocated symbol per access)
  const unsigned kbase = lds0 + i16 * G::ROW + ((g4 ^ G::swz_k(i16)) << 4);
  const int v_row = 4 * g4 + (i16 >> 2);
  const unsigned vbase = lds0 + (VT ? i16 * RV + (((swz_vt(i16)) ^ (g4 >> 1)) << 4) + ((g4 & 1) << 3)  // V^T image: row = d, keys 4 g4 .. of a 32-key step
                                    : v_row * G::ROW + (((((i16 & 3) >> 1)) ^ G::swz_v(v_row)) << 4) + ((i16 & 1) << 3));

  if constexpr ((OX & M16X_PRIO_STATIC) != 0) {
    if (grp == 1) __builtin_amdgcn_s_setprio(1);
  }
  if (grp == 1) {  // group 1 runs one phase behind group 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  stamp(1);
  for (int j = 0; j < T; ++j) {
    const int jn = j + 1 < T ? j + 1 : T - 1;
    const unsigned kb_j = kbase + (j & 1) * G::STAGE, vb_j = vbase + (j & 1) * G::STAGE + G::TILE;
    constexpr bool PAIRED = (OX & M16X_PAIRED_QK) != 0 && NKS == 2 && NKB % 2 == 0 && NOPT % 2 == 0;
    auto kb_of = [](int t) { return PAIRED ? 2 * (t / 4) + (t & 1) : t / NKS; };
    auto ks_of = [](int t) { return PAIRED ? (t >> 1) & 1 : t % NKS; };
    auto k_frag = [&](int t) __attribute__((always_inline)) {
      const int kb = kb_of(t), ks = ks_of(t);
      if constexpr ((OX & M16X_ABL_K) != 0) {
        h8 x = qf[kb % NQB][ks];
        asm volatile("" : "+v"(x));  // opaque: identical MFMAs of different key blocks must not be merged
        return x;
      } else return lds_ld<h8>((kb_j ^ (unsigned)(ks << 6)) + kb * 16 * G::ROW);
    };
    auto v_frag = [&](int idx) __attribute__((always_inline)) {
      const int u = idx / NDB, db = idx % NDB;
      if constexpr ((OX & M16X_ABL_V) != 0) {
        h8 x = qf[db % NQB][u % NKS];
        asm volatile("" : "+v"(x));
        return x;
      } else if constexpr (VT) {
        // A operand row = d = 16 db + i16; k-slots 8 g4 .. + 7 = keys 32u + 4 g4 .. + 3 and 32u + 16 + 4 g4 .. + 3 (the order the P
        // registers have): two plain 8-byte reads 32 bytes apart in the row (chunks 4u + g4/2 and + 2, swizzled by the row)
        return h8_cat(lds_ld<h4>((vb_j ^ (unsigned)((4 * u) << 4)) + (16 * db) * RV), lds_ld<h4>((vb_j ^ (unsigned)((4 * u + 2) << 4)) + (16 * db) * RV));
      } else {
        const unsigned vp = (vb_j ^ (unsigned)(db << 5)) + (32 * u) * G::ROW;
        return h8_cat(lds_read_tr16_at(vp), lds_read_tr16_at(vp + 16 * G::ROW));
      }
    };
    f4 s[NKB][NQB];
    h8 pf[NU][NQB];
    float psum[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) psum[qb] = MS ? -1.0e30f : 0.f;  // MS: running maximum of the exponents instead of the partial row sum
    constexpr bool ONE = (OX & M16X_ONE_STAGE) != 0;
    constexpr int ONE_POS = (OX >> M16X_ONE_POS_SHIFT) & 3;
    auto fetch_whole_tile = [&]() __attribute__((always_inline)) {  // stages = 1: request, wait, (later) use
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G::PPW; ++i) dma_piece(jn, (j + 1) & 1, i);
      hgemm::wait_vmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
    };
    // item it of key block kb: query block it >> 1, registers (it & 1) * 2, + 1 -> k-slots of P^T step kb >> 1
    auto exp_item = [&](int kb, int it, float (&acc)[NQB], bool track = true) __attribute__((always_inline)) {
      const int qb = it >> 1, r = (it & 1) * 2;
      const float x0 = FS ? __builtin_fmaf(s[kb][qb][r], scale_log2e, -m_run[qb]) : s[kb][qb][r];
      const float x1 = FS ? __builtin_fmaf(s[kb][qb][r + 1], scale_log2e, -m_run[qb]) : s[kb][qb][r + 1];
      const float a0 = (OX & M16X_ABL_EXP) != 0 ? x0 : __builtin_amdgcn_exp2f(x0);
      const float a1 = (OX & M16X_ABL_EXP) != 0 ? x1 : __builtin_amdgcn_exp2f(x1);
      if constexpr (MS) {
        if (track) acc[qb] = fmaxf(fmaxf(acc[qb], x0), x1);  // v_max3_f32
      } else if constexpr ((OX & M16X_DOT2_SUM) == 0) acc[qb] += a0 + a1;
      const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
      if constexpr (!MS && (OX & M16X_DOT2_SUM) != 0) acc[qb] = __builtin_amdgcn_fdot2(a, h2{(half_t)1.0f, (half_t)1.0f}, acc[qb], false);
      // an input-only empty asm is a chained node of the instruction selector: the item stays in the step it was
      // written in (without it hipcc sinks every exponential below the last MFMA of the phase)
      asm volatile("" ::"v"(a), "v"(acc[qb]));
      const int u = kb >> 1, e = (kb & 1) * 4 + r;
      pf[u][qb][e] = a[0], pf[u][qb][e + 1] = a[1];
    };

    // ================= phase A: S^T = K Q^T, block kb - 1 exponentiated behind the MFMAs of block kb
    constexpr bool LATE = (OX & M16X_LATE) != 0;
    constexpr int LAG = LATE ? NKB - NOPT : 1;  // block kb - LAG is exponentiated behind the MFMAs of block kb
    if constexpr (ONE && ONE_POS == 0) fetch_whole_tile();
    if constexpr ((OX & M16X_PRIO) != 0 && !LATE) __builtin_amdgcn_s_setprio(1);
    if constexpr ((OX & M16X_PRIO_B) != 0) __builtin_amdgcn_s_setprio(0);
    {
      h8 kf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = k_frag(i);
      constexpr int DSTEP = NQK / G::PPW;
#pragma unroll
      for (int t = 0; t < NQK; ++t) {
        const int kb = kb_of(t), ks = ks_of(t);
        constexpr bool FINE_A = (OX & M16X_FINE) != 0 && !PAIRED;
        if constexpr (LATE && (OX & M16X_PRIO) != 0) {
          if (t == LAG * NKS) __builtin_amdgcn_s_setprio(1);
        }
#pragma unroll
        for (int qi = 0; qi < NQB; ++qi) {
          const int qb = (OX & M16X_SNAKE) != 0 && (t & 1) ? NQB - 1 - qi : qi;
          if (ks == 0) s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t % PD], qf[qb][0], minit[qb], 0, 0, 0);  // chain starts at -m
          else s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t % PD], qf[qb][ks], s[kb][qb], 0, 0, 0);
          cln_mfma_keep(s[kb][qb], kf[t % PD], qf[qb][ks]);  // destination disjoint from the operands (common.h)
          if constexpr (FINE_A) {
            __builtin_amdgcn_sched_barrier(0);
            if (kb >= LAG && kb - LAG < NOPT) {  // item i of the step goes behind MFMA i * NQB / PER_STEP
#pragma unroll
              for (int i = 0; i < PER_STEP; ++i)
                if (i * NQB / PER_STEP == qi && ks * PER_STEP + i < NPAIR) exp_item(kb - LAG, ks * PER_STEP + i, psum);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // the MFMAs of the step first: the VALU slice runs in their shadow
        if (t + PD < NQK) kf[t % PD] = k_frag(t + PD);
        if (!ONE && (OX & M16X_ABL_DMA) == 0 && (t % DSTEP) == DSTEP - 1) dma_piece(jn, (j + 1) & 1, t / DSTEP);
        if constexpr (PAIRED) {
          // group g = t / 4 works on blocks 2g, 2g + 1; the two blocks of group g - 1 are exponentiated over its four steps
          const int g = t / 4, r = t % 4, eb = 2 * (g - 1) + (r >> 1);
          if (g >= 1 && eb < NOPT) {
#pragma unroll
            for (int it = (r & 1) * PER_STEP; it < ((r & 1) + 1) * PER_STEP && it < NPAIR; ++it) exp_item(eb, it, psum);
          }
        } else if (!FINE_A && kb >= LAG && kb - LAG < NOPT) {
#pragma unroll
          for (int it = ks * PER_STEP; it < (ks + 1) * PER_STEP && it < NPAIR; ++it) exp_item(kb - LAG, it, psum);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (ONE && ONE_POS == 1) fetch_whole_tile();
    {
      // ---- the check: partial sums of the optimistic blocks, raw scores of the deferred ones
      bool bad = false;
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        bad |= MS ? !(psum[qb] <= 15.0f) : !(psum[qb] <= 32768.0f);
        if constexpr ((OX & M16X_LATE_CHECK) == 0) {
          float mx = s[NOPT][qb][0];
#pragma unroll
          for (int kb = NOPT; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
          bad |= (FS ? __builtin_fmaf(mx, scale_log2e, -m_run[qb]) : mx) > 14.0f;
        }
      }
      const bool first = j == 0;  // tile 0 has no reference yet: it adopts its true maximum
      if (first || __builtin_amdgcn_ballot_w64(bad) != 0) {
        // ---- cold path: true row maxima, standard rescale, the optimistic blocks again
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
          float mx = s[0][qb][0];
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
          const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
          mx = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
          const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
          float d = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));  // relative to the running reference
          if constexpr (FS) d = __builtin_fmaf(d, scale_log2e, -m_run[qb]);      // (raw maximum -> scaled, relative)
          const float delta = first ? d : fmaxf(d, 0.f);
          const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
          m_run[qb] += delta;
          l_run[qb] *= alpha;
          if constexpr (MS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) lacc[qb][r] *= alpha;
          }
          if constexpr (!FS) {  // (FS: the raw scores stay, exp_item subtracts the new m_run)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
              for (int r = 0; r < 4; ++r) s[kb][qb][r] -= delta;
#pragma unroll
            for (int r = 0; r < 4; ++r) minit[qb][r] = -m_run[qb];
            asm volatile("" : "+v"(minit[qb]));
          }
#pragma unroll
          for (int b = 0; b < NDB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[b][qb][r] *= alpha;
          psum[qb] = MS ? -1.0e30f : 0.f;
        }
#pragma unroll
        for (int kb = 0; kb < NOPT; ++kb)
#pragma unroll
          for (int it = 0; it < NPAIR; ++it) exp_item(kb, it, psum);
      }
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) l_run[qb] += MS ? 0.f : psum[qb];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    if constexpr ((OX & M16X_ABL_BAR) == 0) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ================= phase B: O^T += V^T P^T; the deferred key blocks are exponentiated under its first MFMAs
    if constexpr ((OX & M16X_PRIO) != 0 && !LATE) __builtin_amdgcn_s_setprio(0);
    if constexpr ((OX & M16X_PRIO_B) != 0) __builtin_amdgcn_s_setprio(1);
    if constexpr (ONE && ONE_POS == 2) fetch_whole_tile();
    float psum_d[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) psum_d[qb] = 0.f;
    h8 vf[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) vf[i] = v_frag(i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int idx = 0; idx < NPV; ++idx) {
      const int u = idx / NDB, b = idx % NDB;
      constexpr bool FINE_B = (OX & M16X_FINE) != 0;
      if constexpr ((OX & M16X_LATE_CHECK) != 0) {
        if (idx == (NDEF * NPAIR + DRATE - 1) / DRATE) {  // every deferred exponential is done, none of them has been consumed yet
          bool bad_d = false;
#pragma unroll
          for (int qb = 0; qb < NQB; ++qb) bad_d |= !(psum_d[qb] <= 32768.0f);
          if (__builtin_amdgcn_ballot_w64(bad_d) != 0) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
              float mx = s[NOPT][qb][0];
#pragma unroll
              for (int kb = NOPT; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
              const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
              mx = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
              const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
              const float delta = fmaxf(fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1])), 0.f);
              const float alpha = __builtin_amdgcn_exp2f(-delta);
              m_run[qb] += delta;
              l_run[qb] *= alpha;  // holds this tile's optimistic blocks already
#pragma unroll
              for (int kb = NOPT; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kb][qb][r] -= delta;
#pragma unroll
              for (int r = 0; r < 4; ++r) minit[qb][r] = -m_run[qb];
              asm volatile("" : "+v"(minit[qb]));
#pragma unroll
              for (int bb = 0; bb < NDB; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r) ot[bb][qb][r] *= alpha;
              psum_d[qb] = 0.f;
            }
#pragma unroll
            for (int kb = NOPT; kb < NKB; ++kb)
#pragma unroll
              for (int it = 0; it < NPAIR; ++it) exp_item(kb, it, psum_d);
          }
        }
      }
      if constexpr (LATE && (OX & M16X_PRIO) != 0) {
        if (idx == (NDEF * NPAIR + DRATE - 1) / DRATE) __builtin_amdgcn_s_setprio(0);  // the deferred items are done: bare MFMAs from here
      }
#pragma unroll
      for (int qi = 0; qi < NQB; ++qi) {
        const int qb = (OX & M16X_SNAKE) != 0 && (idx & 1) ? NQB - 1 - qi : qi;
        ot[b][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[idx % PD], pf[u][qb], ot[b][qb], 0, 0, 0);
        cln_mfma_keep(ot[b][qb], vf[idx % PD], pf[u][qb]);
        if constexpr (FINE_B) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < DRATE; ++i) {
            const int it = idx * DRATE + i;
            if (i * NQB / DRATE == qi && it < NDEF * NPAIR) exp_item(NOPT + it / NPAIR, it % NPAIR, psum_d, false);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (MS) {
        if (b == NDB - 1) {  // the last d-block of P^T step u: every P of the step is final -- its row sums, on the matrix pipe
#pragma unroll
          for (int qb = 0; qb < NQB; ++qb) {
            lacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[u][qb], lacc[qb], 0, 0, 0);
            cln_mfma_keep(lacc[qb], ones, pf[u][qb]);
          }
        }
      }
      if (idx + PD < NPV) vf[idx % PD] = v_frag(idx + PD);
      // deferred items at DRATE per step: all of them are done before the first P^T step that holds a deferred block
#pragma unroll
      for (int it = idx * DRATE; !FINE_B && it < (idx + 1) * DRATE && it < NDEF * NPAIR; ++it) exp_item(NOPT + it / NPAIR, it % NPAIR, psum_d, false);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) l_run[qb] += MS ? 0.f : psum_d[qb];
    if constexpr (ONE && ONE_POS == 3) fetch_whole_tile();
    hgemm::wait_vmcnt<0>();  // own DMA pieces of tile j+1 landed
    if constexpr ((OX & M16X_ABL_BAR) == 0) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  stamp(2);
  if (grp == 0) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: O = O^T / l, staged through LDS (wave-private rows). Lane (query 16*qb + i16) holds d = 16*b + 4*g4 .. +3.
  const int lane_e = cln_fresh_lane(), i16_e = lane_e & 15, g4_e = lane_e >> 4;
  char* ob = smem + wave * (G::RPW * G::OS);
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    float l_tot = MS ? lacc[qb][0] : l_run[qb];
    if constexpr (!MS) {
      const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
      l_tot = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
      const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
      l_tot = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
    }
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][qb][e] * inv);
      *reinterpret_cast<h4*>(ob + (qb * 16 + i16_e) * G::OS + (b * 16 + g4_e * 4) * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  constexpr int LPR = D / 8;
  half_t* og = O + head + (size_t)q_row0 * D;
#pragma unroll 4
  for (int it = 0; it < (G::RPW * LPR) / 64; ++it) {
    const int idx = it * 64 + lane_e;
    const int row = idx / LPR, c = idx % LPR;
    cln_store_stream(reinterpret_cast<u4*>(og + (size_t)row * D + c * 8), *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16), (OX & M16X_NT_STORE) != 0 ? 1 : 0);
  }
  if constexpr (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the O rows have left
    stamp(3);
    if (stamps && lane_e == 0) {
      unsigned long long* p = stamps + ((size_t)blockIdx.x * (G::NT / 64) + wave) * 10;
#pragma unroll
      for (int i = 0; i < 4; ++i) p[2 * i] = st_rt[i], p[2 * i + 1] = st_mt[i];
      p[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID: wave / SIMD / CU / SH / SE
      p[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    }
  }
}

template <int D_, int RPW_, int BC_, int PD = 4, int NDEF = 1, int OX = 0, bool VT = false>
int launch_m16x(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoM16<D_, RPW_, BC_>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_m16x_kernel<D_, RPW_, BC_, PD, NDEF, OX, VT>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)G::D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_m16x_kernel<D_, RPW_, BC_, PD, NDEF, OX, VT>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e, (OX & M16X_STAMP) != 0 ? g_m16x_stamps : nullptr);
  return cln_check_launch();
}

}  // namespace fa2
