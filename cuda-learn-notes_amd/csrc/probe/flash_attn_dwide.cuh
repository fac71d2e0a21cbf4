// FlashAttention-2 forward for head dims 768 / 1024: three / four waves split the head dim of one 32-row query group.
// Reference rungs: the fine-grained tiling kernels, whose MAX_HEADDIM table goes up to d = 1024
// (kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qkv.cu:70, :732; flash_attn_mma.py:436-506).
//
// Same idea as the D = 512 kernel of flash_attn_dsplit.cuh, one step further: a QUAD of waves owns 32 query rows,
// wave `part` holds Q[:, part*256 .. +256) (64 registers) and O^T[part*256 .. +256, :] (128 registers), computes a
// partial S^T over its quarter of d, the four partials are summed through LDS (every wave writes 4 KiB and reads
// its three partners': identical sums in all four, fp32 adds in the same order), then each wave does P.V for its own
// 256 output columns. No S recomputation (the previous path recomputed S for each of four output slices: 2.5x the
// MFMA work, 100 TF at [1,16,4096,1024], below torch SDPA).
// 2 groups of D/256 waves = 64 query rows per workgroup (8 waves at D = 1024, 6 at D = 768). One K tile and one V
// tile of 32 keys fill the LDS (D = 1024: 2 x 64 KiB + 32 KiB of exchange = 160 KiB), so there is no ring: V_j is fetched while the workgroup computes
// QK^T of tile j, K_{j+1} while it computes softmax + PV of tile j -- each buffer is refilled during the phase that
// does not read it, two workgroup barriers per tile.
#pragma once
#include "flash_attn_dsplit.cuh"

namespace fa2 {

template <int D>
struct GeoWide {
  static constexpr int NSP = D / 256, BC = 32, NW = 2 * NSP, BR = 64, NT = NW * 64, DH = 256;
  static constexpr int ROW = D * 2;
  static constexpr int TILE = BC * ROW;
  static constexpr int SX = NW * 4096;
  static constexpr int OS = DH * 2 + 16;
  static constexpr int EPI = NW * 32 * OS;
  static constexpr int LDS_BYTES = 2 * TILE + SX;
  static constexpr int PPW = TILE / 1024 / NW;  // DMA pieces per wave per tile (all waves fill both operands)
  static constexpr int PPR = ROW / 1024;        // pieces per row
  static_assert(D == 1024 || D == 768, "d-wide kernel: D = 768 / 1024 (256 columns of d per wave)");
  static_assert(NW * 1024 == 4 * ROW, "one DMA round of the workgroup = 4 rows");
  static_assert(EPI <= LDS_BYTES && LDS_BYTES <= 160 * 1024, "LDS");
};

// PAD (D = 768 only): the tensors have dreal = 640 columns; see flash_attn_dsplit.cuh for the padding rules.
template <int D, int OPT, bool PAD = false>
__global__ __launch_bounds__(GeoWide<D>::NT, 1) void fa2_fwd_dwide_kernel(const half_t* __restrict__ Q,
                                                               const half_t* __restrict__ K,
                                                               const half_t* __restrict__ V, half_t* __restrict__ O,
                                                               int N, int n_qblk, int n_heads, float scale_log2e,
                                                               int dreal) {
  using G = GeoWide<D>;
  const int DR = PAD ? dreal : D;  // columns per row in memory
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int part = wave % G::NSP, rg = wave / G::NSP;

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
  const size_t head = (size_t)head_i * N * DR;
  const int q_row0 = qb * G::BR + rg * 32;
  const unsigned lds0 = hgemm::lds_addr_of(smem);
  const char* Kh = reinterpret_cast<const char*>(K + head);
  const char* Vh = reinterpret_cast<const char*>(V + head);

  // ---- LDS-DMA: wave w fills the 1-KiB pieces i*NW + w of the K tile and of the V tile. One round of the NW waves
  // covers NW KiB = exactly 4 rows, so lane l of wave w always carries chunk c of row 4*i + row0 with
  //   row0 = (w*1024 + l*16) / ROW  (0..3),   c = ((w*1024 + l*16) % ROW) / 16   -- both lane constants.
  // Swizzles as in flash_attn_bigd.cuh (low 4 bits of the chunk index):
  //   K: row & 15 = ((4*i) & 15) + row0  (disjoint bits),      V: (row & 3) << 2 = row0 << 2.
  const int off0 = wave * 1024 + lane * 16, row0 = off0 / G::ROW, c0 = (off0 % G::ROW) >> 4;
  const unsigned k_src_lane = (unsigned)(row0 * G::ROW + ((c0 ^ row0) << 4));
  const unsigned v_src_lane = (unsigned)(row0 * G::ROW + ((c0 ^ (row0 << 2)) << 4));
  auto dma_k = [&](int jt, int i) {
    if constexpr (PAD) {  // memory rows are DR*2 bytes: row base from row0, source chunk clamped into the row
      const unsigned chunk = min((unsigned)((c0 ^ row0) ^ ((4 * i) & 15)), (unsigned)(DR / 8 - 1));
      hgemm::glds16_asm(Kh + (size_t)jt * (G::BC * DR * 2) + i * 4 * (DR * 2), (unsigned)(row0 * DR * 2) + (chunk << 4),
                        lds0 + (i * G::NW + wave) * 1024);
    } else {
      hgemm::glds16_asm(Kh + (size_t)jt * G::TILE + i * 4 * G::ROW, k_src_lane ^ (unsigned)(((4 * i) & 15) << 4),
                        lds0 + (i * G::NW + wave) * 1024);
    }
  };
  auto dma_v = [&](int jt, int i) {
    if constexpr (PAD) {
      const unsigned chunk = min((unsigned)(c0 ^ (row0 << 2)), (unsigned)(DR / 8 - 1));
      hgemm::glds16_asm(Vh + (size_t)jt * (G::BC * DR * 2) + i * 4 * (DR * 2), (unsigned)(row0 * DR * 2) + (chunk << 4),
                        lds0 + G::TILE + (i * G::NW + wave) * 1024);
    } else {
      hgemm::glds16_asm(Vh + (size_t)jt * G::TILE + i * 4 * G::ROW, v_src_lane, lds0 + G::TILE + (i * G::NW + wave) * 1024);
    }
  };

  h8 qf[G::DH / 16];
  {
    const half_t* qp = Q + head + (size_t)(q_row0 + l31) * DR + part * G::DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < G::DH / 16; ++ks) {
      if (!PAD || part * G::DH + ks * 16 < DR) qf[ks] = *reinterpret_cast<const h8*>(qp + ks * 16);
      else qf[ks] = h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  f16v ot[G::DH / 32];
#pragma unroll
  for (int b = 0; b < G::DH / 32; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[b][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const int T = N / G::BC;
  __builtin_assume(T > 0);
#pragma unroll
  for (int i = 0; i < G::PPW; ++i) dma_k(0, i);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), compiler-visible (see flash_attn_bigd.cuh)
#pragma unroll
  for (int ks = 0; ks < G::DH / 16; ++ks) asm volatile("" : "+v"(qf[ks]));  // keep the Q loads out of the KV loop
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int kbase = l31 * G::ROW + ((hi ^ (l31 & 15)) << 4) + part * 512;
  const int i16 = lane & 15;
  const int v_row = 4 * hi + (i16 >> 2);
  const int vbase = G::TILE + v_row * G::ROW + ((((v_row & 3) << 2) + (((lane >> 4) & 1) * 2) + ((i16 & 3) >> 1)) << 4) +
                    ((i16 & 1) << 3) + part * 512;
  auto k_frag = [&](int ks) { return *reinterpret_cast<const h8*>(smem + (kbase ^ ((ks & 7) << 5)) + (ks >> 3) * 256); };
  auto v_frag = [&](int idx) {
    const int st = idx / (G::DH / 32), b = idx % (G::DH / 32);
    const char* vp = smem + (vbase ^ ((b & 3) << 6)) + (16 * st) * G::ROW + (b >> 2) * 256;
    return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::ROW));
  };
  char* sx_mine = smem + 2 * G::TILE + wave * 4096 + lane * 16;
  const char* sx_grp = smem + 2 * G::TILE + rg * G::NSP * 4096 + lane * 16;

  constexpr int NK = G::DH / 16, NPV = 2 * (G::DH / 32);
  for (int j = 0; j < T; ++j) {
    // ================= phase 1: partial S^T over this wave's quarter of d; V_j streams into its buffer
    f16v s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const h8 kf = k_frag(ks);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s, 0, 0, 0);
      cln_mfma_keep(s, kf, qf[ks]);  // destination disjoint from the operands (common.h)
      if ((ks % (NK / G::PPW)) == NK / G::PPW - 1) dma_v(j, ks / (NK / G::PPW));
      if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<f4*>(sx_mine + q * 1024) = f4{s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]};
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) + lgkmcnt(0): own V pieces landed, the partial is in LDS
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ================= phase 2: S = sum of the four partials (same order in every wave), softmax, PV; K_{j+1} streams
    const int jn = j + 1 < T ? j + 1 : T - 1;  // past the end: reload the last tile (branch-free; nobody reads it)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f4 acc = *reinterpret_cast<const f4*>(sx_grp + q * 1024);
#pragma unroll
      for (int p = 1; p < G::NSP; ++p) {
        const f4 t = *reinterpret_cast<const f4*>(sx_grp + p * 4096 + q * 1024);
        acc[0] += t[0], acc[1] += t[1], acc[2] += t[2], acc[3] += t[3];
      }
      s[4 * q] = acc[0], s[4 * q + 1] = acc[1], s[4 * q + 2] = acc[2], s[4 * q + 3] = acc[3];
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
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
      for (int b = 0; b < G::DH / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          float t0 = ot[b][r], t1 = ot[b][r + 1], t2 = ot[b][r + 2], t3 = ot[b][r + 3];
          asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
          ot[b][r] = t0 * alpha, ot[b][r + 1] = t1 * alpha, ot[b][r + 2] = t2 * alpha, ot[b][r + 3] = t3 * alpha;
        }
    }
    h8 pf[2];
    {
      const float nm = -m_run;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float a0 = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, nm));
        const float a1 = __builtin_amdgcn_exp2f(fmaf(s[r + 1], scale_log2e, nm));
        psum += a0 + a1;
        const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
        pf[r >> 3][r & 7] = a[0], pf[r >> 3][(r & 7) + 1] = a[1];
      }
      l_run += psum;
    }
#pragma unroll
    for (int idx = 0; idx < NPV; ++idx) {
      const int st = idx / (G::DH / 32), b = idx % (G::DH / 32);
      const h8 vf = v_frag(idx);
      ot[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[st], ot[b], 0, 0, 0);
      cln_mfma_keep(ot[b], vf, pf[st]);  // destination disjoint from the operands (common.h)
      if ((idx % (NPV / G::PPW)) == NPV / G::PPW - 1) dma_k(jn, idx / (NPV / G::PPW));
      if ((idx & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    hgemm::wait_vmcnt<0>();  // own K pieces landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: O = O^T / l, staged through LDS (wave-private rows)
  float l_tot;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = 1.0f / l_tot;
  char* ob = smem + wave * (32 * G::OS);
#pragma unroll
  for (int b = 0; b < G::DH / 32; ++b) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][rq * 4 + e] * inv);
      *reinterpret_cast<h4*>(ob + l31 * G::OS + (b * 32 + rq * 8 + hi * 4) * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  constexpr int LPR = G::DH / 8;
  half_t* og = O + head + (size_t)q_row0 * DR + part * G::DH;
#pragma unroll 4
  for (int it = 0; it < (32 * LPR) / 64; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / LPR, c = idx % LPR;
    const u4 v = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
    if (!PAD || part * G::DH + c * 8 < DR) *reinterpret_cast<u4*>(og + (size_t)row * DR + c * 8) = v;
  }
}

template <int D, int OPT, bool PAD = false>
int launch_dwide(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream,
                 int dreal = D) {
  using G = GeoWide<D>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  if (PAD ? (dreal % 64 != 0 || dreal <= D - 256 || dreal >= D) : dreal != D) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_dwide_kernel<D, OPT, PAD>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)dreal);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_dwide_kernel<D, OPT, PAD>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e, dreal);
  return cln_check_launch();
}

}  // namespace fa2
