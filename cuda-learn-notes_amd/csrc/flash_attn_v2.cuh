// FlashAttention-2 forward, second-generation gfx950 kernel (head dims 32..256).
//
// Same mathematics and register-level dataflow as probe/flash_attn.cuh (swapped S^T = K Q^T so a lane
// owns one query row; P^T stays in registers as the B operand of O^T = V^T P^T; V fragments fetched
// with ds_read_b64_tr_b16 in the kv order of the P registers), re-structured around what the first
// GPU profile showed was missing:
//   * NW = 8 waves x 32 query rows per workgroup (Br = 256) or NW = 4 (Br = 128, two workgroups per
//     CU): one K/V staging pass feeds twice the MFMA work;
//   * K/V tiles DOUBLE-buffered in LDS with ONE workgroup barrier per KV tile: global loads of tile
//     j+2 are issued right after tile j+1's registers were written to LDS and have a full iteration
//     to land (issue-early / write-late, cdna guide T14);
//   * softmax with v_max3 chains, one v_permlane32_swap for the cross-half exchange (no LDS
//     bpermute), exp2 with the 1/sqrt(d)*log2(e) scale folded into one v_fma, packed RNE conversion
//     v_cvt_pk_f16_f32, and a deferred running-max update (rescale O only when some row's max grew by
//     more than 2^8; cdna guide T13 -- the previous tile's P V is always complete before the decision,
//     and l is updated with the same factor);
//   * s_setprio(1) around the MFMA clusters;
//   * epilogue staged through LDS so O leaves as full 16-byte-per-lane row segments;
//   * heads pinned to XCDs (block id -> (head, q-block) remap) so a head's K/V stays in one L2.
// Reference functions replaced: kernels/flash-attn/mma/basic/flash_attn_mma_share_qkv.cu:66 (config C4),
// share_kv.cu:66, split_q.cu:52, and their swizzle/ acc_f32 siblings.
#pragma once
#include "common.h"

namespace fa2 {

enum : int { OPT_DEFER = 1, OPT_PRIO = 2, OPT_LDS_EPI = 4, OPT_XCD = 8, OPT_STAGGER = 16, OPT_ONES = 32, OPT_SOLO = 64, OPT_PK = 512, OPT_VPRE = 1024, OPT_KPRE = 2048, OPT_SOFTEXP = 4096, OPT_SOFTEXP_HALF = 8192, OPT_PRE = 16384, OPT_PD8 = 32768, OPT_PD16 = 65536, OPT_SUMM = 131072, OPT_1STAGE = 262144, OPT_ABL_K = 1 << 19, OPT_ABL_V = 1 << 20, OPT_ABL_XR = 1 << 21, OPT_ABL_XW = 1 << 22, OPT_ABL_DMA = 1 << 23, OPT_ABL_BAR = 1 << 24, OPT_SPREAD = 1 << 25, OPT_DEFAULT = 15 };  // OPT_ABL_*: probe ablations of the ring kernel (results are garbage by design)
enum : int { ABL_NO_SOFTMAX = 1, ABL_NO_STAGE = 2, ABL_NO_FRAG_READS = 4, ABL_NO_BARRIER = 8 };

// exp2 on the plain VALU (experiment): ubench/overlap.hip shows v_exp_f32 does not overlap with the SIMD partner's
// MFMA stream while v_fma_f32 does. Round-to-nearest split x = n + r, degree-3 minimax of 2^r on [-0.5, 0.5]
// (max rel. error 7.5e-5, below the fp16 rounding of P), exponent added with one v_lshl_add_u32.
__device__ __forceinline__ float soft_exp2(float x) {
  x = fmaxf(x, -126.0f);
  const float magic = 12582912.0f;  // 1.5 * 2^23: low mantissa bits of (x + magic) hold round(x)
  const float t = x + magic;
  const float r = x - (t - magic);
  const float p = fmaf(fmaf(fmaf(0.0551716481f, r, 0.242611121f), r, 0.693260989f), r, 0.999928074f);
  return __uint_as_float(__float_as_uint(p) + (__float_as_uint(t) << 23));
}

template <int D, int NW, bool VT>
struct Geo {
  static constexpr int BC = 64, BR = NW * 32, NT = NW * 64;
  static constexpr int KS = D * 2 + 16;  // K row stride: +16 B makes ds_read_b128 over 32 rows conflict-free
  static constexpr int VPAD = ((D * 2) % 128 == 64) ? 0 : 64;
  static constexpr int VS = VT ? (BC * 2 + 8) : (D * 2 + VPAD);  // see probe/flash_attn.cuh
  static constexpr int K_BYTES = BC * KS;
  static constexpr int V_BYTES = VT ? D * VS : BC * VS;
  static constexpr int STAGE = (K_BYTES + V_BYTES + 15) / 16 * 16;
  static constexpr int OS = D * 2 + 16;  // epilogue row stride
  static constexpr int EPI_BYTES = NW * 32 * OS;
  static constexpr int LDS_BYTES = (2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
  static constexpr int CHUNKS = BC * (D / 8);        // 16-byte chunks per K (or V) tile
  static constexpr int CH = (CHUNKS + NT - 1) / NT;  // per thread
  static constexpr bool EXACT = (CHUNKS % NT) == 0;
  static_assert(D % 32 == 0 && D <= 256, "head dim");
  static_assert(D <= 128 || NW <= 4, "D > 128 needs the whole register file: one wave per SIMD");
};

// (D = 128 with 2-wave workgroups does not fit 256 registers without spilling: it takes the one-wave-per-SIMD budget)
template <int D, int NW, bool VT, int OPT, int ABL>
__global__ __launch_bounds__(NW * 64, ((D > 128 || (D >= 96 && NW == 2)) ? 1 : 2)) void fa2_fwd_v2_kernel(const half_t* __restrict__ Q,
                                                                const half_t* __restrict__ K,
                                                                const half_t* __restrict__ V,
                                                                half_t* __restrict__ O, int N, int n_qblk,
                                                                int n_heads, float scale_log2e) {
  using G = Geo<D, NW, VT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  // ---- block id -> (head, q-block). Hardware places block b on XCD b % 8: give every XCD whole heads.
  int head_i, qb;
  {
    const int bid = blockIdx.x;
    if ((OPT & OPT_XCD) && (n_heads & 7) == 0) {
      const int xcd = bid & 7, k = bid >> 3;
      head_i = (k / n_qblk) * 8 + xcd;
      qb = k - (k / n_qblk) * n_qblk;
    } else {
      head_i = bid / n_qblk;
      qb = bid - head_i * n_qblk;
    }
  }
  const size_t head = (size_t)head_i * N * D;
  const int q_row0 = qb * G::BR + wave * 32;
  const half_t* Kh = K + head;
  const half_t* Vh = V + head;

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q = l31) holds d = 16*ks + 8*hi .. +7
  h8 qf[D / 16];
  {
    const half_t* qp = Q + head + (size_t)(q_row0 + l31) * D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) qf[ks] = *reinterpret_cast<const h8*>(qp + ks * 16);
  }

  // ---- staging (global -> VGPR -> LDS)
  u4 kreg[G::CH], vreg[G::CH];
  auto load_tile = [&](int j) {
#pragma unroll
    for (int u = 0; u < G::CH; ++u) {
      const int idx = tid + u * G::NT;
      if (G::EXACT || idx < G::CHUNKS) {
        const int row = idx / (D / 8), ch = idx % (D / 8);
        kreg[u] = *reinterpret_cast<const u4*>(Kh + (size_t)(j * 64 + row) * D + ch * 8);
        if constexpr (VT) {
          const int vrow = idx >> 3, vch = idx & 7;  // V^T tile: D rows x 64 kv = D*8 chunks
          vreg[u] = *reinterpret_cast<const u4*>(Vh + (size_t)vrow * N + j * 64 + vch * 8);
        } else {
          vreg[u] = *reinterpret_cast<const u4*>(Vh + (size_t)(j * 64 + row) * D + ch * 8);
        }
      }
    }
  };
  auto write_tile = [&](int buf) {
    char* kb = smem + buf * G::STAGE;
    char* vb = kb + G::K_BYTES;
#pragma unroll
    for (int u = 0; u < G::CH; ++u) {
      const int idx = tid + u * G::NT;
      if (G::EXACT || idx < G::CHUNKS) {
        const int row = idx / (D / 8), ch = idx % (D / 8);
        *reinterpret_cast<u4*>(kb + row * G::KS + ch * 16) = kreg[u];
        if constexpr (VT) {
          const int vrow = idx >> 3, vch = idx & 7;
          char* p = vb + vrow * G::VS + vch * 16;  // 8-byte aligned only (VS = 136): two b64 stores
          *reinterpret_cast<u2*>(p) = u2{vreg[u][0], vreg[u][1]};
          *reinterpret_cast<u2*>(p + 8) = u2{vreg[u][2], vreg[u][3]};
        } else {
          *reinterpret_cast<u4*>(vb + row * G::VS + ch * 16) = vreg[u];
        }
      }
    }
  };

  // ---- accumulators
  f16v ot[D / 32];  // O^T[d-block]: column = own q row, rows = d
#pragma unroll
  for (int b = 0; b < D / 32; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[b][r] = 0.f;
  f16v lacc;  // OPT_ONES: row sums from the matrix pipe (A = ones): every row of the result is sum_kv P^T[kv][q]
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
  const h8 ones = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
  // OPT_PRE (see flash_attn_dsplit.cuh): Q pre-multiplied by log2(e)/sqrt(d), both S^T accumulators of a tile start at
  // -m through the C operand of their first MFMA: P = exp2(acc), no per-score v_fma, no accumulator zeroing
  constexpr bool QPRE = (OPT & OPT_PRE) != 0;
  float m_run = QPRE ? 0.f : -1.0e30f;  // running max, scaled log2 domain (finite sentinel: no inf arithmetic)
  float l_run = 0.f;       // per-lane partial row sum (this lane's half of each kv tile)
  f16v minit;
#pragma unroll
  for (int r = 0; r < 16; ++r) minit[r] = 0.f;
  if constexpr (QPRE) asm volatile("" : "+v"(minit));

  // lane-constant LDS offsets
  const int k_off = l31 * G::KS + hi * 16;
  int v_off;
  if constexpr (VT) {
    v_off = l31 * G::VS + (4 * hi) * 2;
  } else {
    const int i = lane & 15;
    v_off = ((i >> 2) + 4 * hi) * G::VS + (((lane >> 4) & 1) * 16 + (i & 3) * 4) * 2;
  }

  if constexpr ((OPT & OPT_STAGGER) != 0) {  // experiment: offset the two co-resident workgroups of a CU
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_sleep(12);
  }
  const int T = N / 64;
  load_tile(0);
  write_tile(0);
  // Everything issued so far (Q fragments, tile 0) must be COMPLETE in hipcc's scoreboard before the loop: the Q
  // loads are otherwise first waited for inside the loop body (`s_waitcnt vmcnt(0)` ahead of the first MFMA that
  // reads qf[2..]), and because vmcnt retires in order that wait also drains the K/V prefetch issued at the end
  // of the previous iteration -- the full global-load latency was exposed once per KV tile.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), lgkmcnt/expcnt untouched
  if constexpr (QPRE) {
    const half_t sc = (half_t)scale_log2e;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) qf[ks] = qf[ks] * sc;
  }
  // OPT_1STAGE (the names' stages = 1 on the small-grid shapes): tile j + 1 is requested at the END of tile j and waited for there
  // (load, wait, write to LDS, barrier, use): no global load of the wave is in flight while it computes; same arithmetic, same bits
  constexpr bool ONE_STAGE = (OPT & OPT_1STAGE) != 0;
  if (!ONE_STAGE && T > 1) load_tile(1);
  __syncthreads();

  h8 kf_const[2], vf_const;  // ABL_NO_FRAG_READS only
  if constexpr ((ABL & ABL_NO_FRAG_READS) != 0) {
    kf_const[0] = *reinterpret_cast<const h8*>(smem + k_off);
    kf_const[1] = *reinterpret_cast<const h8*>(smem + k_off + 32);
    vf_const = *reinterpret_cast<const h8*>(smem + G::K_BYTES + l31 * 16);
  }

  for (int j = 0; j < T; ++j) {
    const char* kb = smem + (j & 1) * G::STAGE;
    const char* vb = kb + G::K_BYTES;

    // ---- S^T = K Q^T : two 32-kv sub-tiles
    f16v s0, s1;
    if constexpr (QPRE) {
      s0 = minit, s1 = minit;  // consumed as the C operand of the first MFMA pair: no moves
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[r] = 0.f, s1[r] = 0.f;
    }
    if constexpr ((OPT & OPT_KPRE) != 0) {
      // all K fragment reads of this tile up front (in groups of 4 k-steps): the LDS latency is paid once per
      // group instead of once per MFMA pair (hipcc otherwise issues each read two MFMAs ahead of its use)
      constexpr int GRP = 4;
#pragma unroll
      for (int g0 = 0; g0 < D / 16; g0 += GRP) {
        h8 ka[GRP], kc[GRP];
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
          ka[u] = *reinterpret_cast<const h8*>(kb + k_off + (g0 + u) * 32);
          kc[u] = *reinterpret_cast<const h8*>(kb + k_off + 32 * G::KS + (g0 + u) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
          s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[u], qf[g0 + u], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kc[u], qf[g0 + u], s1, 0, 0, 0);
          cln_mfma_keep(s0, ka[u], qf[g0 + u]);  // destinations disjoint from the operands (common.h)
          cln_mfma_keep(s1, kc[u], qf[g0 + u]);
        }
      }
    } else {
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      h8 kf0, kf1;
      if constexpr ((ABL & ABL_NO_FRAG_READS) != 0) {
        kf0 = kf_const[0], kf1 = kf_const[1];
      } else {
        kf0 = *reinterpret_cast<const h8*>(kb + k_off + ks * 32);
        kf1 = *reinterpret_cast<const h8*>(kb + k_off + 32 * G::KS + ks * 32);
      }
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qf[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qf[ks], s1, 0, 0, 0);
      cln_mfma_keep(s0, kf0, qf[ks]);
      cln_mfma_keep(s1, kf1, qf[ks]);
    }
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(0);
    }
    // V fragments of the first PRE k-steps are fetched BEFORE the softmax (V_j is already resident): they land
    // while the VALU works
    constexpr int PRE = (OPT & OPT_VPRE) ? (D <= 64 ? 4 : (D <= 128 ? 2 : 0)) : 0;
    h8 vpre[PRE > 0 ? PRE : 1][D / 32];
    if constexpr (PRE > 0 && !VT) {
#pragma unroll
      for (int st = 0; st < PRE; ++st)
#pragma unroll
        for (int b = 0; b < D / 32; ++b) {
          const char* vp = vb + v_off + (32 * (st >> 1) + 16 * (st & 1)) * G::VS + b * 64;
          vpre[st][b] = h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::VS));
        }
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- online softmax (lane-local row)
    h8 pf[4];  // P^T fragments of the four 16-kv k-steps, in accumulator register order
    if constexpr ((ABL & ABL_NO_SOFTMAX) != 0) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const h2 a = __builtin_convertvector(f2{s0[r], s0[r + 1]}, h2);
        const h2 b = __builtin_convertvector(f2{s1[r], s1[r + 1]}, h2);
        pf[r >> 3][r & 7] = a[0], pf[r >> 3][(r & 7) + 1] = a[1];
        pf[2 + (r >> 3)][r & 7] = b[0], pf[2 + (r >> 3)][(r & 7) + 1] = b[1];
      }
      l_run += 1.f;
    } else {
      float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      const float dgrow = QPRE ? mx : mx * scale_log2e - m_run;  // growth of the row max (log2 domain)
      bool grow;
      if constexpr ((OPT & OPT_DEFER) != 0) grow = dgrow > 8.0f;
      else grow = dgrow > 0.f;
      const bool first = QPRE && j == 0;  // tile 0 adopts its max unconditionally (the accumulators started at 0)
      if (first || __builtin_amdgcn_ballot_w64(grow) != 0) {  // wave-uniform: some row's max moved
        // (without OPT_PRE m_run starts at the -1e30 sentinel: take the max directly, "m_run + delta" would cancel)
        const float delta = first ? dgrow : fmaxf(dgrow, 0.f);
        const float m_new = QPRE ? m_run + delta : fmaxf(m_run, mx * scale_log2e);
        const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
        if constexpr (QPRE) {  // the pending scores were accumulated from the old -m
#pragma unroll
          for (int r = 0; r < 16; ++r) s0[r] -= delta, s1[r] -= delta;
#pragma unroll
          for (int r = 0; r < 16; ++r) minit[r] = -m_run;
          asm volatile("" : "+v"(minit));
        }
        if constexpr ((OPT & OPT_ONES) != 0) lacc[0] *= alpha;
#pragma unroll
        for (int b = 0; b < D / 32; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[b][r] *= alpha;
      }
      const float nm = -m_run;
      float psum = 0.f;
      if constexpr ((OPT & OPT_PK) != 0) {
        // packed-f32 VALU: one v_pk_fma_f32 / v_pk_add_f32 does two lanes-values per 4-cycle issue slot
        // (PMC: a plain VALU op and a packed one both hold the VALU for one quad-cycle; v_exp_f32 two)
        const f2 c2 = {scale_log2e, scale_log2e}, nm2 = {nm, nm};
        f2 ps = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f2 ea = __builtin_elementwise_fma(f2{s0[r], s0[r + 1]}, c2, nm2);
          const f2 eb = __builtin_elementwise_fma(f2{s1[r], s1[r + 1]}, c2, nm2);
          const f2 pa = {__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
          const f2 pb = {__builtin_amdgcn_exp2f(eb[0]), __builtin_amdgcn_exp2f(eb[1])};
          ps += pa;
          ps += pb;
          const h2 a = __builtin_convertvector(pa, h2);
          const h2 b = __builtin_convertvector(pb, h2);
          pf[r >> 3][r & 7] = a[0], pf[r >> 3][(r & 7) + 1] = a[1];
          pf[2 + (r >> 3)][r & 7] = b[0], pf[2 + (r >> 3)][(r & 7) + 1] = b[1];
        }
        psum = ps[0] + ps[1];
      } else {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float a0, a1, b0, b1;
        if constexpr ((OPT & OPT_SOFTEXP) != 0) {
          a0 = soft_exp2(fmaf(s0[r], scale_log2e, nm));
          a1 = soft_exp2(fmaf(s0[r + 1], scale_log2e, nm));
          b0 = soft_exp2(fmaf(s1[r], scale_log2e, nm));
          b1 = soft_exp2(fmaf(s1[r + 1], scale_log2e, nm));
        } else if constexpr ((OPT & OPT_SOFTEXP_HALF) != 0) {
          a0 = soft_exp2(fmaf(s0[r], scale_log2e, nm));
          a1 = soft_exp2(fmaf(s0[r + 1], scale_log2e, nm));
          b0 = __builtin_amdgcn_exp2f(fmaf(s1[r], scale_log2e, nm));
          b1 = __builtin_amdgcn_exp2f(fmaf(s1[r + 1], scale_log2e, nm));
        } else {
          a0 = __builtin_amdgcn_exp2f(QPRE ? s0[r] : fmaf(s0[r], scale_log2e, nm));
          a1 = __builtin_amdgcn_exp2f(QPRE ? s0[r + 1] : fmaf(s0[r + 1], scale_log2e, nm));
          b0 = __builtin_amdgcn_exp2f(QPRE ? s1[r] : fmaf(s1[r], scale_log2e, nm));
          b1 = __builtin_amdgcn_exp2f(QPRE ? s1[r + 1] : fmaf(s1[r + 1], scale_log2e, nm));
        }
        if constexpr ((OPT & OPT_ONES) == 0) psum += (a0 + a1) + (b0 + b1);
        const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
        const h2 b = __builtin_convertvector(f2{b0, b1}, h2);
        pf[r >> 3][r & 7] = a[0], pf[r >> 3][(r & 7) + 1] = a[1];
        pf[2 + (r >> 3)][r & 7] = b[0], pf[2 + (r >> 3)][(r & 7) + 1] = b[1];
      }
      }
      l_run += psum;
    }

    // ---- O^T += V^T P^T
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      // kv rows of this lane's P registers in k-step st: base + {0..3} and base + 8 + {0..3}, base = kv0 + 4*hi
      const int kv0 = 32 * (st >> 1) + 16 * (st & 1);
      if constexpr ((OPT & OPT_ONES) != 0) lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf[st], lacc, 0, 0, 0);
#pragma unroll
      for (int b = 0; b < D / 32; ++b) {
        h8 vf;
        if constexpr ((ABL & ABL_NO_FRAG_READS) != 0) {
          vf = vf_const;
        } else if constexpr (PRE > 0 && !VT) {
          if (st < PRE) {
            vf = vpre[st][b];
          } else {
            const char* vp = vb + v_off + kv0 * G::VS + b * 64;
            vf = h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::VS));
          }
        } else if constexpr (VT) {
          const char* vp = vb + v_off + b * 32 * G::VS + kv0 * 2;
          vf = h8_cat(*reinterpret_cast<const h4*>(vp), *reinterpret_cast<const h4*>(vp + 16));
        } else {
          const char* vp = vb + v_off + kv0 * G::VS + b * 64;
          vf = h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::VS));
        }
        ot[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[st], ot[b], 0, 0, 0);
        cln_mfma_keep(ot[b], vf, pf[st]);
      }
    }
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(0);

    // ---- stage: tile j+1 registers -> the other buffer (its readers finished before the last barrier),
    //      then issue tile j+2's global loads (a full iteration to land)
    if constexpr ((ABL & ABL_NO_STAGE) == 0) {
      if constexpr (ONE_STAGE) {
        if (j + 1 < T) {
          load_tile(j + 1);
          __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
          write_tile((j + 1) & 1);
        }
      } else {
        if (j + 1 < T) write_tile((j + 1) & 1);
        if (j + 2 < T) load_tile(j + 2);
      }
    }
    if constexpr ((ABL & ABL_NO_BARRIER) == 0) __syncthreads();
  }

  // ---- epilogue: O = O^T / l
  float l_tot;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  if constexpr ((OPT & OPT_ONES) != 0) l_tot = lacc[0];
  const float inv = 1.0f / l_tot;
  if constexpr ((OPT & OPT_LDS_EPI) != 0) {
    // the loop's last barrier guarantees every wave is done with the K/V buffers
    const int lane_e = cln_fresh_lane(), l31 = lane_e & 31, hi = lane_e >> 5;  // not carried across the KV loop
    const int lane = lane_e;
    char* ob = smem + wave * (32 * G::OS);
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][rq * 4 + e] * inv);
        *reinterpret_cast<h4*>(ob + l31 * G::OS + (b * 32 + rq * 8 + hi * 4) * 2) = o;
      }
    // wave-private region, LDS ops of one wave complete in order: no barrier
    constexpr int LPR = D / 8;  // 16-byte chunks per row
    half_t* og = O + head + (size_t)q_row0 * D;
#pragma unroll
    for (int it = 0; it < (32 * LPR + 63) / 64; ++it) {
      const int idx = it * 64 + lane;
      if ((32 * LPR) % 64 == 0 || idx < 32 * LPR) {
        const int row = idx / LPR, c = idx % LPR;
        const u4 v = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
        *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = v;
      }
    }
  } else {
    half_t* op = O + head + (size_t)(q_row0 + l31) * D;
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][rq * 4 + e] * inv);
        *reinterpret_cast<h4*>(op + b * 32 + rq * 8 + hi * 4) = o;
      }
  }
}

template <int D, int NW, bool VT, int OPT, int ABL = 0>
int launch_v2(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = Geo<D, NW, VT>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  // OPT_SOLO (experiment): over-ask LDS so only one workgroup fits a CU
  constexpr int LDS = ((OPT & OPT_SOLO) != 0 && G::LDS_BYTES < 96 * 1024) ? 96 * 1024 : G::LDS_BYTES;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (LDS > 48 * 1024 && cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_v2_kernel<D, NW, VT, OPT, ABL>), LDS) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_v2_kernel<D, NW, VT, OPT, ABL>), dim3(n_qblk * B * H), dim3(G::NT), LDS, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
