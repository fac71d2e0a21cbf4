// FlashAttention-2 forward, head dims 64 / 128: the sum-checked optimistic-softmax kernel of flash_attn_m16x.cuh as a
// ONE-WAVE-PER-SIMD stream (round 3 probe). Reference rung: kernels/flash-attn/mma/basic/flash_attn_mma_share_qkv.cu:66.
//
// 4 waves x 64 query rows = 256 rows per workgroup (config C4 still puts a workgroup on every CU), each wave alone on its
// SIMD with the whole 512-entry register file. Against the 8-wave kernel: every K / V fragment read from LDS feeds FOUR
// 16x16x32 MFMAs instead of two (half the fragment bytes per flop -- the largest single cost in the energy ablation of
// round 2), one barrier per KV tile instead of two, no partner wave: the exponentials have to hide under the wave's OWN
// MFMAs. That is what the optimistic softmax makes possible without a max pass: block kb - 1 is exponentiated behind the
// MFMA chain of block kb in phase A, and the blocks deferred into phase B are spread over the PV steps at the lowest
// uniform rate that meets every P^T k-step's deadline (item group g -- the two key blocks of k-step NOPT/2 + g -- must be
// complete when PV step (NOPT/2 + g) * NDB starts).
// Register classes are chosen by hand: with the builtin hipcc parks the S^T accumulators in the AGPR half of the 512-entry
// file, which no VALU instruction can read -- one v_accvgpr_read per score (889 of them in the 128-key form), i.e. the VALU
// budget the design lives on. So the MFMAs are inline asm: S^T tiles in VGPRs ("=&v": early-clobber, the destination is
// never an operand's register -- common.h cln_mfma_keep), O^T tiles tied to AGPRs ("+a"). Inline asm is invisible to hipcc's
// hazard pass; the MFMA-result -> VALU-read distances are kept by construction (block kb - 1 is read >= 4 MFMAs after its
// last MFMA) and by an s_nop pad in front of the check and of the epilogue.
// LDS: 2-slot ring of (K tile | V tile), every wave fetches its quarter of both by LDS-DMA during phase A of the tile before.
#pragma once
#include "flash_attn_m16x.cuh"

namespace fa2 {

template <int D_, int BC_>
struct GeoM16S {
  using G8 = GeoM16<D_, 64, BC_>;  // swizzles, row geometry and step counts of the 8-wave form with 64 rows per wave
  static constexpr int D = D_, RPW = 64, BC = BC_, NW = 4, BR = RPW * NW, NT = 256;
  static constexpr int ROW = G8::ROW, TILE = G8::TILE, STAGE = G8::STAGE, RING = G8::RING;
  static constexpr int OS = G8::OS, EPI = NW * RPW * OS;
  static constexpr int LDS_BYTES = RING > EPI ? RING : EPI;
  static constexpr int PPW = G8::PPW;  // pieces per wave per OPERAND tile: a wave issues 2 * PPW per KV tile
};

template <int D_, int BC_, int PD = 4, int NDEF = 2, bool FINE = true>
__global__ __launch_bounds__(256, 1) void fa2_fwd_m16s_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                              const half_t* __restrict__ V, half_t* __restrict__ O,
                                                              int N, int n_qblk, int n_heads, float scale_log2e) {
  using GS = GeoM16S<D_, BC_>;
  using G = typename GS::G8;
  constexpr int D = G::D, NKB = G::NKB, NKS = G::NKS, NQB = G::NQB, NU = G::NU, NDB = G::NDB, NQK = G::NQK, NPV = G::NPV;
  constexpr int NOPT = NKB - NDEF;   // key blocks exponentiated in phase A
  constexpr int NPAIR = NQB * 2;     // (query block, register pair) items of one key block
  constexpr int PER_STEP = (NPAIR + NKS - 1) / NKS;
  static_assert(NDEF >= 1 && NOPT >= 2, "at least one whole P^T k-step is complete at the end of phase A");
  // lowest uniform rate (items per PV step) that meets every deferred block's deadline: block NOPT + i belongs to P^T
  // k-step (NOPT + i) / 2, whose first PV step is ((NOPT + i) / 2) * NDB
  constexpr int rate_of = [] {
    int r = 1;
    for (int i = 0; i < NDEF; ++i) {
      const int need = (i + 1) * NPAIR, steps = ((NOPT + i) / 2) * NDB;
      const int q = (need + steps - 1) / steps;
      if (q > r) r = q;
    }
    return r;
  }();
  constexpr int DRATE = rate_of;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g4 = lane >> 4;

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
  const int q_row0 = qb_i * GS::BR + wave * GS::RPW;
  const unsigned lds0 = hgemm::lds_addr_of(smem);

  const char* k_h = reinterpret_cast<const char*>(K + head);
  const char* v_h = reinterpret_cast<const char*>(V + head);
  const int lr = lane / G::CPR, lc = lane % G::CPR;
  const unsigned k_lane = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ G::swz_k(wave * G::RPP + lr)) << 4);
  const unsigned v_lane = (unsigned)(lr * G::ROW) + (unsigned)((lc ^ G::swz_v(wave * G::RPP + lr)) << 4);
  // piece p of a KV tile: p < PPW -> K piece p*4 + wave, else V piece (p - PPW)*4 + wave
  auto dma_piece = [&](int jt, int slot, int p) __attribute__((always_inline)) {
    const bool is_v = p >= G::PPW;
    const int piece = (is_v ? p - G::PPW : p) * 4 + wave;
    const char* s = (is_v ? v_h : k_h) + (size_t)jt * G::TILE + piece * 1024;
    hgemm::glds16_asm(s, is_v ? v_lane : k_lane, lds0 + slot * G::STAGE + (is_v ? G::TILE : 0) + piece * 1024);
  };

  h8 qf[NQB][NKS];
  const int T = N / G::BC;
  __builtin_assume(T > 0);
#pragma unroll
  for (int p = 0; p < 2 * G::PPW; ++p) dma_piece(0, 0, p);
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
    for (int qb = 0; qb < NQB; ++qb) {
      ot[b][qb] = f4{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+a"(ot[b][qb]));
    }
  float m_run[NQB], l_run[NQB];
  f4 minit[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    m_run[qb] = 0.f, l_run[qb] = 0.f;
    minit[qb] = f4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(minit[qb]));
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible: also retires the Q loads
  {
    const half_t sc = (half_t)scale_log2e;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        qf[qb][ks] = qf[qb][ks] * sc;
        asm volatile("" : "+v"(qf[qb][ks]));
      }
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int kbase = i16 * G::ROW + ((g4 ^ G::swz_k(i16)) << 4);
  const int v_row = 4 * g4 + (i16 >> 2);
  const int vbase = v_row * G::ROW + (((((i16 & 3) >> 1)) ^ G::swz_v(v_row)) << 4) + ((i16 & 1) << 3);

  for (int j = 0; j < T; ++j) {
    const int jn = j + 1 < T ? j + 1 : T - 1;
    const int kb_j = kbase + (j & 1) * G::STAGE, vb_j = vbase + (j & 1) * G::STAGE + G::TILE;
    auto k_frag = [&](int t) __attribute__((always_inline)) {
      const int kb = t / NKS, ks = t % NKS;
      return *reinterpret_cast<const h8*>(smem + (kb_j ^ (ks << 6)) + kb * 16 * G::ROW);
    };
    auto v_frag = [&](int idx) __attribute__((always_inline)) {
      const int u = idx / NDB, db = idx % NDB;
      const char* vp = smem + (vb_j ^ (db << 5)) + (32 * u) * G::ROW;
      return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 16 * G::ROW));
    };
    f4 s[NKB][NQB];
    h8 pf[NU][NQB];
    float psum[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) psum[qb] = 0.f;
    auto exp_item = [&](int kb, int it, float (&acc)[NQB]) __attribute__((always_inline)) {
      const int qb = it >> 1, r = (it & 1) * 2;
      const float a0 = __builtin_amdgcn_exp2f(s[kb][qb][r]);
      const float a1 = __builtin_amdgcn_exp2f(s[kb][qb][r + 1]);
      acc[qb] += a0 + a1;
      const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
      asm volatile("" ::"v"(a), "v"(acc[qb]));  // the item stays in the step it was written in (see flash_attn_m16x.cuh)
      const int u = kb >> 1, e = (kb & 1) * 4 + r;
      pf[u][qb][e] = a[0], pf[u][qb][e + 1] = a[1];
    };

    // ================= phase A: S^T = K Q^T, block kb - 1 exponentiated behind the MFMAs of block kb; the wave's pieces of
    // tile j + 1 (K, then V) go out one every DSTEP steps
    {
      h8 kf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = k_frag(i);
      constexpr int NPIECE = 2 * G::PPW;
      constexpr int DSTEP = NQK / NPIECE > 0 ? NQK / NPIECE : 1;
      static_assert(NQK >= NPIECE, "one DMA piece per step at most");
#pragma unroll
      for (int t = 0; t < NQK; ++t) {
        const int kb = t / NKS, ks = t % NKS;
        // ONE MFMA, then its share of the step's VALU work: an in-order wave cannot issue VALU instructions behind an MFMA
        // that is itself waiting for the matrix pipe, so four MFMAs back to back leave the pipe idle while the VALU slice runs
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
          if (ks == 0)  // chain starts at -m
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(s[kb][qb]) : "v"(kf[t % PD]), "v"(qf[qb][0]), "v"(minit[qb]));
          else
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(s[kb][qb]) : "v"(kf[t % PD]), "v"(qf[qb][ks]));
          if (FINE) {
            if (qb == NQB - 1) {
              if (t + PD < NQK) kf[t % PD] = k_frag(t + PD);
              if ((t % DSTEP) == DSTEP - 1 && t / DSTEP < NPIECE) dma_piece(jn, (j + 1) & 1, t / DSTEP);
            }
            if (kb >= 1 && kb - 1 < NOPT) {
#pragma unroll
              for (int it = ks * PER_STEP + qb; it < (ks + 1) * PER_STEP && it < NPAIR; it += NQB) exp_item(kb - 1, it, psum);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (!FINE) {
          __builtin_amdgcn_sched_barrier(0);
          if (t + PD < NQK) kf[t % PD] = k_frag(t + PD);
          if ((t % DSTEP) == DSTEP - 1 && t / DSTEP < NPIECE) dma_piece(jn, (j + 1) & 1, t / DSTEP);
          if (kb >= 1 && kb - 1 < NOPT) {
#pragma unroll
            for (int it = ks * PER_STEP; it < (ks + 1) * PER_STEP && it < NPAIR; ++it) exp_item(kb - 1, it, psum);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // first V fragments: their latency runs under the check. The pad: the last QK^T MFMAs' results are read by the check.
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    h8 vf[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) vf[i] = v_frag(i);
    {
      // ---- the check: partial sums of the optimistic blocks, raw scores of the deferred ones
      bool bad = false;
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        bad |= !(psum[qb] <= 32768.0f);
        float mx = s[NOPT][qb][0];
#pragma unroll
        for (int kb = NOPT; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
        bad |= mx > 14.0f;
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
          const float d = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));  // relative to the running reference
          const float delta = first ? d : fmaxf(d, 0.f);
          const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
          m_run[qb] += delta;
          l_run[qb] *= alpha;
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][qb][r] -= delta;
#pragma unroll
          for (int r = 0; r < 4; ++r) minit[qb][r] = -m_run[qb];
          asm volatile("" : "+v"(minit[qb]));
#pragma unroll
          for (int b = 0; b < NDB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[b][qb][r] *= alpha;
          psum[qb] = 0.f;
        }
#pragma unroll
        for (int kb = 0; kb < NOPT; ++kb)
#pragma unroll
          for (int it = 0; it < NPAIR; ++it) exp_item(kb, it, psum);
      }
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) l_run[qb] += psum[qb];
    }

    // ================= phase B: O^T += V^T P^T; the deferred key blocks are exponentiated under its MFMAs at DRATE per step
    float psum_d[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) psum_d[qb] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int idx = 0; idx < NPV; ++idx) {
      const int u = idx / NDB, b = idx % NDB;
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ot[b][qb]) : "v"(vf[idx % PD]), "v"(pf[u][qb]));
        if (FINE) {
          if (qb == NQB - 1 && idx + PD < NPV) vf[idx % PD] = v_frag(idx + PD);
#pragma unroll
          for (int it = idx * DRATE + qb; it < (idx + 1) * DRATE && it < NDEF * NPAIR; it += NQB) exp_item(NOPT + it / NPAIR, it % NPAIR, psum_d);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!FINE) {
        __builtin_amdgcn_sched_barrier(0);
        if (idx + PD < NPV) vf[idx % PD] = v_frag(idx + PD);
#pragma unroll
        for (int it = idx * DRATE; it < (idx + 1) * DRATE && it < NDEF * NPAIR; ++it) exp_item(NOPT + it / NPAIR, it % NPAIR, psum_d);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) l_run[qb] += psum_d[qb];
    hgemm::wait_vmcnt<0>();  // own DMA pieces of tile j+1 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: O = O^T / l, staged through LDS (wave-private rows). Lane (query 16*qb + i16) holds d = 16*b + 4*g4 .. +3.
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // last PV MFMA -> v_accvgpr_read (hazard pass cannot see the asm MFMAs)
#pragma unroll
  for (int b = 0; b < NDB; ++b)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) asm volatile("" : "+a"(ot[b][qb]));
  const int lane_e = cln_fresh_lane(), i16_e = lane_e & 15, g4_e = lane_e >> 4;
  char* ob = smem + wave * (GS::RPW * GS::OS);
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    float l_tot = l_run[qb];
    {
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
      *reinterpret_cast<h4*>(ob + (qb * 16 + i16_e) * GS::OS + (b * 16 + g4_e * 4) * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  constexpr int LPR = D / 8;
  half_t* og = O + head + (size_t)q_row0 * D;
#pragma unroll 4
  for (int it = 0; it < (GS::RPW * LPR) / 64; ++it) {
    const int idx = it * 64 + lane_e;
    const int row = idx / LPR, c = idx % LPR;
    *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = *reinterpret_cast<const u4*>(ob + row * GS::OS + c * 16);
  }
}

template <int D_, int BC_, int PD = 4, int NDEF = 2, bool FINE = true>
int launch_m16s(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using GS = GeoM16S<D_, BC_>;
  if (N % GS::BR != 0 || N % BC_ != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_m16s_kernel<D_, BC_, PD, NDEF, FINE>), GS::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)GS::D);
  const int n_qblk = N / GS::BR;
  CLN_LAUNCH((fa2_fwd_m16s_kernel<D_, BC_, PD, NDEF, FINE>), dim3(n_qblk * B * H), dim3(GS::NT), GS::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
