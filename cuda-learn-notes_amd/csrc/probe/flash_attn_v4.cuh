// FlashAttention-2 forward, three-phase ping-pong variant of the v2 kernel (same Geo, fragment maps and
// softmax as flash_attn_v2.cuh; 8 waves x 32 query rows).
//
// Measured on v2 (profiles/r01_pmc_fa_v2_variants.json): the two waves of a SIMD run QK^T -> softmax -> PV in
// lockstep (one barrier per KV tile keeps them aligned), so both contend for the matrix pipe, then both for the
// VALU: MFMA-busy 47 %, VALU-active 41-73 %, the rest is waiting. v4 keeps every wave's instruction stream as it
// is but runs the two 4-wave groups ONE PHASE APART, three barriers per tile (the HGEMM ping-pong idea applied
// to attention; cdna guide "two waves per SIMD" items 1-3):
//
//   slot:      3j+1          3j+2          3j+3
//   group A:   QK^T(j)       softmax(j)    P V(j)
//   group B:   softmax(j)    P V(j)        QK^T(j+1)          (B starts one slot early with QK^T(0))
//
// so in every slot one resident of each SIMD is on the matrix pipe while the other is on the VALU (slot 3j+3 is
// matrix || matrix: unavoidable with three phases and two groups).
// Staging rides in the softmax phase of each group (its own half of the tile): registers holding tile j+1 are
// written to LDS, then tile j+2's global loads are issued. Buffer safety (2 K + 2 V buffers):
//   K_{j+1} overwrites K_{j-1}, last read in slot 3j-2 (A's QK^T(j-1)); written in slots 3j+1 (B) and 3j+2 (A);
//   first read in slot 3j+3 (B's QK^T(j+1)).
//   V_{j+1} overwrites V_{j-1}, last read in slot 3j (A's P V(j-1)); written in slots 3j+1 / 3j+2; first read in
//   slot 3j+5. Every hand-over crosses at least one workgroup barrier that follows the writes' lgkmcnt(0).
#pragma once
#include "flash_attn_v2.cuh"

namespace fa2 {

template <int D, bool VT, int OPT>
__global__ __launch_bounds__(512, 2) void fa2_fwd_v4_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                            const half_t* __restrict__ V, half_t* __restrict__ O,
                                                            int N, int n_qblk, int n_heads, float scale_log2e) {
  constexpr int NW = 8;
  using G = Geo<D, NW, VT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2;  // 0 = A (waves 0-3), 1 = B (waves 4-7): the two residents of every SIMD
  const int l31 = lane & 31, hi = lane >> 5;

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

  h8 qf[D / 16];
  {
    const half_t* qp = Q + head + (size_t)(q_row0 + l31) * D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) qf[ks] = *reinterpret_cast<const h8*>(qp + ks * 16);
  }

  u4 kreg[G::CH], vreg[G::CH];
  auto load_tile = [&](int j) {
#pragma unroll
    for (int u = 0; u < G::CH; ++u) {
      const int idx = tid + u * G::NT;
      if (G::EXACT || idx < G::CHUNKS) {
        const int row = idx / (D / 8), ch = idx % (D / 8);
        kreg[u] = *reinterpret_cast<const u4*>(Kh + (size_t)(j * 64 + row) * D + ch * 8);
        if constexpr (VT) {
          const int vrow = idx >> 3, vch = idx & 7;
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
          char* p = vb + vrow * G::VS + vch * 16;
          *reinterpret_cast<u2*>(p) = u2{vreg[u][0], vreg[u][1]};
          *reinterpret_cast<u2*>(p + 8) = u2{vreg[u][2], vreg[u][3]};
        } else {
          *reinterpret_cast<u4*>(vb + row * G::VS + ch * 16) = vreg[u];
        }
      }
    }
  };

  f16v ot[D / 32];
#pragma unroll
  for (int b = 0; b < D / 32; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[b][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;
  const int k_off = l31 * G::KS + hi * 16;
  int v_off;
  if constexpr (VT) {
    v_off = l31 * G::VS + (4 * hi) * 2;
  } else {
    const int i = lane & 15;
    v_off = ((i >> 2) + 4 * hi) * G::VS + (((lane >> 4) & 1) * 16 + (i & 3) * 4) * 2;
  }

  f16v s0, s1;
  h8 pf[4];
  const int T = N / 64;

  auto phase_qk = [&](int j) {
    const char* kb = smem + (j & 1) * G::STAGE;
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = 0.f, s1[r] = 0.f;
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      const h8 kf0 = *reinterpret_cast<const h8*>(kb + k_off + ks * 32);
      const h8 kf1 = *reinterpret_cast<const h8*>(kb + k_off + 32 * G::KS + ks * 32);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qf[ks], s0, 0, 0, 0);
      cln_mfma_keep(s0, kf0, qf[ks]);  // destination disjoint from the operands (common.h)
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qf[ks], s1, 0, 0, 0);
      cln_mfma_keep(s1, kf1, qf[ks]);  // destination disjoint from the operands (common.h)
    }
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(0);
  };
  auto phase_sm = [&](int j) {
    // staging first: tile j+1 registers -> LDS, then tile j+2 loads (in flight for a whole tile period)
    if (j + 1 < T) write_tile((j + 1) & 1);
    if (j + 2 < T) load_tile(j + 2);
    float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const float mxs = mx * scale_log2e;
    bool grow;
    if constexpr ((OPT & OPT_DEFER) != 0) grow = (mxs - m_run) > 8.0f;
    else grow = mxs > m_run;
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {
      const float m_new = fmaxf(m_run, mxs);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int b = 0; b < D / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[b][r] *= alpha;
    }
    const float nm = -m_run;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float a0 = __builtin_amdgcn_exp2f(fmaf(s0[r], scale_log2e, nm));
      const float a1 = __builtin_amdgcn_exp2f(fmaf(s0[r + 1], scale_log2e, nm));
      const float b0 = __builtin_amdgcn_exp2f(fmaf(s1[r], scale_log2e, nm));
      const float b1 = __builtin_amdgcn_exp2f(fmaf(s1[r + 1], scale_log2e, nm));
      psum += (a0 + a1) + (b0 + b1);
      const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
      const h2 b = __builtin_convertvector(f2{b0, b1}, h2);
      pf[r >> 3][r & 7] = a[0], pf[r >> 3][(r & 7) + 1] = a[1];
      pf[2 + (r >> 3)][r & 7] = b[0], pf[2 + (r >> 3)][(r & 7) + 1] = b[1];
    }
    l_run += psum;
  };
  auto phase_pv = [&](int j) {
    const char* vb = smem + (j & 1) * G::STAGE + G::K_BYTES;
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int kv0 = 32 * (st >> 1) + 16 * (st & 1);
#pragma unroll
      for (int b = 0; b < D / 32; ++b) {
        h8 vf;
        if constexpr (VT) {
          const char* vp = vb + v_off + b * 32 * G::VS + kv0 * 2;
          vf = h8_cat(*reinterpret_cast<const h4*>(vp), *reinterpret_cast<const h4*>(vp + 16));
        } else {
          const char* vp = vb + v_off + kv0 * G::VS + b * 64;
          vf = h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::VS));
        }
        ot[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[st], ot[b], 0, 0, 0);
        cln_mfma_keep(ot[b], vf, pf[st]);  // destination disjoint from the operands (common.h)
      }
    }
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(0);
  };

  // prologue: tile 0 resident, tile 1 in registers
  load_tile(0);
  write_tile(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // Q + tile 0 complete in hipcc's scoreboard (see flash_attn_v2.cuh)
  if (T > 1) load_tile(1);
  __syncthreads();

  if (group == 0) {
    __syncthreads();  // slot 0: group B alone does QK^T(0)
    for (int j = 0; j < T; ++j) {
      phase_qk(j);
      __syncthreads();
      phase_sm(j);
      __syncthreads();
      phase_pv(j);
      __syncthreads();
    }
  } else {
    phase_qk(0);
    __syncthreads();
    for (int j = 0; j < T; ++j) {
      phase_sm(j);
      __syncthreads();
      phase_pv(j);
      __syncthreads();
      if (j + 1 < T) phase_qk(j + 1);
      __syncthreads();
    }
  }

  // ---- epilogue
  float l_tot;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = 1.0f / l_tot;
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
  constexpr int LPR = D / 8;
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
}

template <int D, bool VT, int OPT>
int launch_v4(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = Geo<D, 8, VT>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (G::LDS_BYTES > 48 * 1024 && cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_v4_kernel<D, VT, OPT>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_v4_kernel<D, VT, OPT>), dim3(n_qblk * B * H), dim3(512), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
