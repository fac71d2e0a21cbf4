// FlashAttention-2 forward, software-pipelined variant of the v2 kernel (same Geo, same fragment maps).
//
// v2 runs QK^T -> softmax -> PV strictly in sequence inside a wave; the two waves of a SIMD execute the same
// phases in lockstep (one barrier per KV tile keeps them aligned), so the matrix pipe idles during both
// waves' softmax and the VALU idles during both waves' MFMA clusters (measured: a single wave per SIMD
// reaches 74 % of the two-wave throughput; removing the softmax VALU work alone gives +32 %).
// v3 overlaps them INSIDE the wave (cdna guide T15 "two P tiles live", here two S tiles):
//
//   step j:   B   S_{j+1} = K_{j+1} Q^T   (MFMA)   ||  P_j = exp2(S_j*c - m), row sums, f16 pack   (VALU)
//             C   O += V_j P_j             (MFMA)   ||  row max of S_{j+1}                           (VALU)
//             D   deferred-max decision for tile j+1 (rare wave-uniform branch: O *= alpha, l *= alpha)
//             stage K_{j+2}, V_{j+1} registers -> LDS, issue loads of K_{j+3}, V_{j+2};  one barrier
//
// B and C are single basic blocks so the machine scheduler can interleave the independent MFMA and VALU
// streams; OPT_SGB pins the interleave with sched_group_barrier. K runs one tile ahead of V: two K buffers
// and two V buffers. The loop is unrolled by two with the S registers swapping roles (static register
// names, cdna guide rule 20); the last tile is peeled (no next QK^T).
// T13 hazard: the rescale decision for tile j+1 is taken after ALL of tile j's P V MFMAs are issued and
// before any P_{j+1} exists; l and O are scaled by the same alpha.
#pragma once
#include <type_traits>
#include "flash_attn_v2.cuh"

namespace fa2 {

enum : int { OPT_SGB = 256, OPT_SGB2 = 16384 };

template <int D, int NW, bool VT, int OPT>
__global__ __launch_bounds__(NW * 64, (D > 128 ? 1 : 2)) void fa2_fwd_v3_kernel(const half_t* __restrict__ Q,
                                                                               const half_t* __restrict__ K,
                                                                               const half_t* __restrict__ V,
                                                                               half_t* __restrict__ O, int N,
                                                                               int n_qblk, int n_heads,
                                                                               float scale_log2e) {
  using G = Geo<D, NW, VT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  constexpr int KB = (G::K_BYTES + 15) / 16 * 16, VB = (G::V_BYTES + 15) / 16 * 16;
  char* const k_ring = smem;           // 2 x KB
  char* const v_ring = smem + 2 * KB;  // 2 x VB

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
  auto load_k = [&](int j) {
#pragma unroll
    for (int u = 0; u < G::CH; ++u) {
      const int idx = tid + u * G::NT;
      if (G::EXACT || idx < G::CHUNKS) {
        const int row = idx / (D / 8), ch = idx % (D / 8);
        kreg[u] = *reinterpret_cast<const u4*>(Kh + (size_t)(j * 64 + row) * D + ch * 8);
      }
    }
  };
  auto load_v = [&](int j) {
#pragma unroll
    for (int u = 0; u < G::CH; ++u) {
      const int idx = tid + u * G::NT;
      if (G::EXACT || idx < G::CHUNKS) {
        if constexpr (VT) {
          const int vrow = idx >> 3, vch = idx & 7;
          vreg[u] = *reinterpret_cast<const u4*>(Vh + (size_t)vrow * N + j * 64 + vch * 8);
        } else {
          const int row = idx / (D / 8), ch = idx % (D / 8);
          vreg[u] = *reinterpret_cast<const u4*>(Vh + (size_t)(j * 64 + row) * D + ch * 8);
        }
      }
    }
  };
  auto write_k = [&](int buf) {
    char* kb = k_ring + buf * KB;
#pragma unroll
    for (int u = 0; u < G::CH; ++u) {
      const int idx = tid + u * G::NT;
      if (G::EXACT || idx < G::CHUNKS) {
        const int row = idx / (D / 8), ch = idx % (D / 8);
        *reinterpret_cast<u4*>(kb + row * G::KS + ch * 16) = kreg[u];
      }
    }
  };
  auto write_v = [&](int buf) {
    char* vb = v_ring + buf * VB;
#pragma unroll
    for (int u = 0; u < G::CH; ++u) {
      const int idx = tid + u * G::NT;
      if (G::EXACT || idx < G::CHUNKS) {
        if constexpr (VT) {
          const int vrow = idx >> 3, vch = idx & 7;
          char* p = vb + vrow * G::VS + vch * 16;
          *reinterpret_cast<u2*>(p) = u2{vreg[u][0], vreg[u][1]};
          *reinterpret_cast<u2*>(p + 8) = u2{vreg[u][2], vreg[u][3]};
        } else {
          const int row = idx / (D / 8), ch = idx % (D / 8);
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

  // S^T tile = K_j Q^T into (a, b)
  auto qk = [&](f16v& a, f16v& b, const char* kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f, b[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      const h8 kf0 = *reinterpret_cast<const h8*>(kb + k_off + ks * 32);
      const h8 kf1 = *reinterpret_cast<const h8*>(kb + k_off + 32 * G::KS + ks * 32);
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qf[ks], a, 0, 0, 0);
      cln_mfma_keep(a, kf0, qf[ks]);  // destination disjoint from the operands (common.h)
      b = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qf[ks], b, 0, 0, 0);
      cln_mfma_keep(b, kf1, qf[ks]);  // destination disjoint from the operands (common.h)
    }
  };
  auto rowmax = [&](const f16v& a, const f16v& b) {
    float mx = fmaxf(a[0], b[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, a[r]), b[r]);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  };
  // deferred running-max update for the tile whose row max is mx (must precede that tile's exp)
  auto decide = [&](float mx) {
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
  };

  // one pipeline step; (c0,c1) = S_j complete, (n0,n1) receives S_{j+1} when NEXT
  auto step = [&](f16v& c0, f16v& c1, f16v& n0, f16v& n1, int j, auto next_tag) {
    constexpr bool NEXT = decltype(next_tag)::value;
    const char* kb_next = k_ring + ((j + 1) & 1) * KB;
    const char* vb = v_ring + (j & 1) * VB;
    // ---- B: next QK^T (MFMA) || exp / sums / pack of the current tile (VALU)
    h8 kpre[(OPT & OPT_SGB2) ? D / 8 : 1];
    if constexpr (NEXT && (OPT & OPT_SGB2) != 0) {
      // all K fragments of the next tile first (D/8 reads in flight together), THEN the MFMA chain with the
      // softmax VALU pinned between consecutive MFMAs: no MFMA waits on a read issued just ahead of it
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        kpre[2 * ks] = *reinterpret_cast<const h8*>(kb_next + k_off + ks * 32);
        kpre[2 * ks + 1] = *reinterpret_cast<const h8*>(kb_next + k_off + 32 * G::KS + ks * 32);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) n0[r] = 0.f, n1[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kpre[2 * ks], qf[ks], n0, 0, 0, 0);
        cln_mfma_keep(n0, kpre[2 * ks], qf[ks]);  // destination disjoint from the operands (common.h)
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kpre[2 * ks + 1], qf[ks], n1, 0, 0, 0);
        cln_mfma_keep(n1, kpre[2 * ks + 1], qf[ks]);  // destination disjoint from the operands (common.h)
      }
    } else if constexpr (NEXT) {
      qk(n0, n1, kb_next);
    }
    h8 pf[4];
    {
      const float nm = -m_run;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float a0 = __builtin_amdgcn_exp2f(fmaf(c0[r], scale_log2e, nm));
        const float a1 = __builtin_amdgcn_exp2f(fmaf(c0[r + 1], scale_log2e, nm));
        const float b0 = __builtin_amdgcn_exp2f(fmaf(c1[r], scale_log2e, nm));
        const float b1 = __builtin_amdgcn_exp2f(fmaf(c1[r + 1], scale_log2e, nm));
        psum += (a0 + a1) + (b0 + b1);
        const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
        const h2 b = __builtin_convertvector(f2{b0, b1}, h2);
        pf[r >> 3][r & 7] = a[0], pf[r >> 3][(r & 7) + 1] = a[1];
        pf[2 + (r >> 3)][r & 7] = b[0], pf[2 + (r >> 3)][(r & 7) + 1] = b[1];
      }
      l_run += psum;
    }
    if constexpr (NEXT && (OPT & OPT_SGB2) != 0) {
      constexpr int NM = D / 8;
      __builtin_amdgcn_sched_group_barrier(0x100, NM, 0);  // every K fragment read up front
#pragma unroll
      for (int g = 0; g < NM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, (96 + NM - 1) / NM, 0);   // plain VALU
        __builtin_amdgcn_sched_group_barrier(0x400, (32 + NM - 1) / NM, 0);   // transcendental
      }
    }
    if constexpr (NEXT && (OPT & OPT_SGB) != 0) {
      // D/8 MFMAs in this block; spread the 32 transcendental + ~80 plain VALU ops and the D/8 fragment reads
      // evenly behind them
      constexpr int NM = D / 8;
#pragma unroll
      for (int g = 0; g < NM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // DS read
        __builtin_amdgcn_sched_group_barrier(0x002, (80 + NM - 1) / NM, 0);  // VALU
        __builtin_amdgcn_sched_group_barrier(0x400, (32 + NM - 1) / NM, 0);  // TRANS
      }
    }
    // ---- C: P V (MFMA) || row max of the next tile (VALU)
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
    float mx_next = 0.f;
    if constexpr (NEXT) mx_next = rowmax(n0, n1);
    // ---- D: decision for tile j+1
    if constexpr (NEXT) decide(mx_next);
    // ---- staging + the one barrier of this tile
    const int T = N / 64;
    if (j + 2 < T) write_k(j & 1);        // K_{j+2} -> the buffer K_j was read from one step ago
    if (j + 1 < T) write_v((j + 1) & 1);  // V_{j+1} -> the buffer V_{j-1} was read from one step ago
    if (j + 3 < T) load_k(j + 3);
    if (j + 2 < T) load_v(j + 2);
    if constexpr (NEXT) __syncthreads();
  };

  const int T = N / 64;
  load_k(0);
  load_v(0);
  write_k(0);
  write_v(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // Q + tile 0 complete in hipcc's scoreboard (see flash_attn_v2.cuh)
  if (T > 1) {
    load_k(1);
    write_k(1);
    load_v(1);
  }
  if (T > 2) load_k(2);
  __syncthreads();

  f16v sa0, sa1, sb0, sb1;
  qk(sa0, sa1, k_ring);
  decide(rowmax(sa0, sa1));
  __syncthreads();  // step 0 restages K_2 over K_0: every wave must be done with its S_0 = K_0 Q^T reads
#pragma unroll
  for (int r = 0; r < 16; ++r) sb0[r] = 0.f, sb1[r] = 0.f;

  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;
  int j = 0;
  for (; j + 2 < T; j += 2) {
    step(sa0, sa1, sb0, sb1, j, yes{});
    step(sb0, sb1, sa0, sa1, j + 1, yes{});
  }
  if (T - j == 2) {
    step(sa0, sa1, sb0, sb1, j, yes{});
    step(sb0, sb1, sa0, sa1, j + 1, no{});
  } else {
    step(sa0, sa1, sb0, sb1, j, no{});
  }

  // ---- epilogue: O = O^T / l
  float l_tot;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = 1.0f / l_tot;
  __syncthreads();  // every wave is done with the K/V rings (the peeled last step has no barrier)
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

template <int D, int NW, bool VT, int OPT>
int launch_v3(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = Geo<D, NW, VT>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  constexpr int KB = (G::K_BYTES + 15) / 16 * 16, VB = (G::V_BYTES + 15) / 16 * 16;
  constexpr int RING = 2 * KB + 2 * VB;
  constexpr int LDS = RING > G::EPI_BYTES ? RING : G::EPI_BYTES;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (LDS > 48 * 1024 && cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_v3_kernel<D, NW, VT, OPT>), LDS) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_v3_kernel<D, NW, VT, OPT>), dim3(n_qblk * B * H), dim3(G::NT), LDS, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
