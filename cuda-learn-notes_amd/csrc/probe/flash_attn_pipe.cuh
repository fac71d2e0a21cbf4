// FlashAttention-2 forward, software-pipelined form of the two-group ping-pong kernel (flash_attn_dsplit.cuh, NSP = 1).
// Reference rungs: the split-Q family (kernels/flash-attn/mma/basic/flash_attn_mma_share_qkv.cu:66, share_kv.cu:66,
// tiling_qk.cu:72) -- same math, same [B,H,N,D] fp16 tensors.
//
// Why: tools/ubench/overlap2.hip and interleave.hip (profiles/r01_mfma_valu_overlap2_ubench.log,
// r01_mfma_valu_interleave_ubench.log): on one SIMD a wave's MFMA stream and its PARTNER wave's VALU stream do not
// overlap at all (time = sum), but ~5 plain VALU instructions issued by the SAME wave right behind each of its own
// v_mfma_f32_32x32x16_f16 are free. So the softmax must be interleaved instruction by instruction with the wave's own
// MFMAs: here the exponentials / conversions / row sums of tile j ride in the issue shadow of the QK^T MFMAs of tile
// j+1 (S double-buffered in registers: 16 x BCB more registers per lane).
//   phase A_j: row max of S_j + (rare) rescale, then [S_{j+1} = K_{j+1} Q^T  ||  P_j = exp2(S_j * scale - m)]
//              group 0 issues the LDS-DMA of K_{j+2}, group 1 of V_{j+1}
//   phase B_j: O^T += V_j^T P_j^T
// Two workgroup barriers per tile, the two 4-wave groups one phase apart, as in the non-pipelined kernel: memory
// traffic of one group still overlaps with the other group's matrix work.
#pragma once
#include "flash_attn_dsplit.cuh"

namespace fa2 {

template <int D, int BCB, int OPT, int ABL = 0>
__global__ __launch_bounds__(512, 1) void fa2_fwd_pipe_kernel(const half_t* __restrict__ Q,
                                                              const half_t* __restrict__ K,
                                                              const half_t* __restrict__ V, half_t* __restrict__ O,
                                                              int N, int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoSplit<D, 1, BCB>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int grp = wave >> 2, widx = wave & 3;

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
  const unsigned lds0 = hgemm::lds_addr_of(smem);

  // LDS-DMA (see flash_attn_dsplit.cuh): group 0 fills K tiles, group 1 V tiles; tile t lives in ring slot t & 1
  const char* src_h = reinterpret_cast<const char*>((grp == 0 ? K : V) + head);
  const int lr = lane / G::CPR, lc = lane % G::CPR, rlow = widx * G::RPP + lr;
  const unsigned src_lane = (unsigned)(lr * G::ROW) + (grp == 0 ? (unsigned)((lc ^ G::swz_k(rlow)) << 4) : (unsigned)((lc ^ G::swz_v(rlow)) << 4));
  const unsigned kmask = grp == 0 ? 0xFFu : 0u;
  auto dma_piece = [&](int jt, int i) {
    const int piece = i * 4 + widx;
    const char* s = src_h + (size_t)jt * G::TILE + piece * 1024;
    hgemm::glds16_asm(s, src_lane ^ ((unsigned)(((i * 4 * G::RPP) & 15) << 4) & kmask),
                      lds0 + (jt & 1) * G::STAGE + grp * G::TILE + piece * 1024);
  };

  h8 qf[D / 16];
  {
    const half_t* qp = Q + head + (size_t)(q_row0 + l31) * D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) qf[ks] = *reinterpret_cast<const h8*>(qp + ks * 16);
  }
  f16v ot[D / 32];
#pragma unroll
  for (int b = 0; b < D / 32; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[b][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const int T = N / G::BC;
#pragma unroll
  for (int i = 0; i < G::PPW; ++i) dma_piece(0, i);  // K_0 | V_0
  if (grp == 0) {
#pragma unroll
    for (int i = 0; i < G::PPW; ++i) dma_piece(T > 1 ? 1 : 0, i);  // K_1 (K runs one tile ahead of V)
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) asm volatile("" : "+v"(qf[ks]));  // keep the Q loads out of the KV loop
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int kbase = l31 * G::ROW + ((hi ^ G::swz_k(l31)) << 4);
  const int i16 = lane & 15;
  const int v_row = 4 * hi + (i16 >> 2);
  const int vbase = v_row * G::ROW + ((G::swz_v(v_row) + (((lane >> 4) & 1) * 2) + ((i16 & 3) >> 1)) << 4) +
                    ((i16 & 1) << 3);

  constexpr int NK = D / 16, NQK = BCB * NK, NPV = 2 * BCB * (D / 32), NE = 16 * BCB;  // NE: S values per lane
  constexpr int PD = 4;  // LDS fragments in flight ahead of their MFMA
  auto k_frag = [&](int tile, int t) {  // keys (t % BCB)*32 + l31, k-step t / BCB of K tile `tile`
    const int ks = t / BCB;
    const int base = kbase + (tile & 1) * G::STAGE;
    return *reinterpret_cast<const h8*>(smem + (base ^ ((ks & 7) << 5)) + (ks >> 3) * 256 + (t % BCB) * 32 * G::ROW);
  };
  auto v_frag = [&](int tile, int idx) {
    const int st = idx / (D / 32), b = idx % (D / 32);
    const int base = vbase + (tile & 1) * G::STAGE + G::TILE;
    const char* vp = smem + (base ^ ((b & 3) << 6)) + (16 * st) * G::ROW + (b >> 2) * 256;
    return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::ROW));
  };

  // ---- S_0 = K_0 Q^T (nothing to overlap with yet)
  f16v s_a[BCB], s_b[BCB];
#pragma unroll
  for (int kb = 0; kb < BCB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s_a[kb][r] = 0.f;
#pragma unroll
  for (int t = 0; t < NQK; ++t) {
    const h8 kf = k_frag(0, t);
    s_a[t % BCB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[t / BCB], s_a[t % BCB], 0, 0, 0);
    cln_mfma_keep(s_a[t % BCB], kf, qf[t / BCB]);  // destination disjoint from the operands (common.h)
    if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }

  __builtin_amdgcn_s_barrier();  // everybody is done with K_0 before group 0 refills its slot in A_0
  asm volatile("" ::: "memory");
  if (grp == 1) {  // group 1 runs one phase behind group 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // one KV tile; the S buffers swap roles every tile (T is even: N % 256 == 0 and BC <= 128), so the loop is
  // unrolled by two instead of copying S_{j+1} into S_j
  auto tile_step = [&](const int j, f16v (&s_cur)[BCB], f16v (&s_nxt)[BCB]) {
    // ================= phase A_j
    // (1) row max of S_j, deferred rescale (uniform branch: kept outside the interleaved block)
    {
      float mx = s_cur[0][0];
#pragma unroll
      for (int kb = 0; kb < BCB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_cur[kb][r]);
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
          for (int r = 0; r < 16; r += 4) {
            float t0 = ot[b][r], t1 = ot[b][r + 1], t2 = ot[b][r + 2], t3 = ot[b][r + 3];
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
            ot[b][r] = t0 * alpha, ot[b][r + 1] = t1 * alpha, ot[b][r + 2] = t2 * alpha, ot[b][r + 3] = t3 * alpha;
          }
      }
    }
    // (2) S_{j+1} = K_{j+1} Q^T with P_j = exp2(S_j * scale - m) in the shadow of the MFMAs: MFMA t is followed by
    //     the exponentials of S values [t*NE/NQK, (t+1)*NE/NQK); every group is fenced so hipcc keeps the order
    h8 pf[2 * BCB];
    {
      const float nm = -m_run;
      float psum = 0.f;
      const bool more = j + 1 < T;  // last tile: no S_{j+1}; the MFMAs run on tile T-1 again and are dropped
      const int tk = more ? j + 1 : j;
      const int jdma = grp == 0 ? j + 2 : j + 1;
      const int jd = jdma < T ? jdma : T - 1;  // past the end: refill a dead slot (branch-free)
#pragma unroll
      for (int kb = 0; kb < BCB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_nxt[kb][r] = 0.f;
      h8 kf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = k_frag(tk, i);
      constexpr int EPM = NE / NQK;  // S values handled behind each MFMA (2 at D = 128 / 64-key tiles, 1 at D = 256)
      static_assert(NE % NQK == 0 && (EPM == 1 || EPM % 2 == 0), "exponentials per MFMA");
#pragma unroll
      for (int t = 0; t < NQK; ++t) {
        if (!(ABL & 16))
          s_nxt[t % BCB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t % PD], qf[t / BCB], s_nxt[t % BCB], 0, 0, 0);
          cln_mfma_keep(s_nxt[t % BCB], kf[t % PD], qf[t / BCB]);  // destination disjoint from the operands (common.h)
        if (t + PD < NQK) kf[t % PD] = k_frag(tk, t + PD);
        if (!(ABL & 1) && (t % (NQK / G::PPW)) == NQK / G::PPW - 1) dma_piece(jd, t / (NQK / G::PPW));
        // softmax slice: elements e0 .. e0 + EPM - 1 of the flattened (kb, r) index; pairs feed one packed conversion
        if constexpr (EPM >= 2) {
#pragma unroll
          for (int e = t * EPM; e < (t + 1) * EPM; e += 2) {
            const int kb = e / 16, r = e % 16, u = e / 8;
            const float a0 = (ABL & 2) ? s_cur[kb][r] : __builtin_amdgcn_exp2f(fmaf(s_cur[kb][r], scale_log2e, nm));
            const float a1 = (ABL & 2) ? s_cur[kb][r + 1] : __builtin_amdgcn_exp2f(fmaf(s_cur[kb][r + 1], scale_log2e, nm));
            psum += a0 + a1;
            const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
            pf[u][e % 8] = a[0], pf[u][e % 8 + 1] = a[1];
          }
        } else {  // one value per MFMA: convert in pairs behind every second MFMA
          if (t & 1) {
            const int e = t - 1, kb = e / 16, r = e % 16, u = e / 8;
            const float a0 = (ABL & 2) ? s_cur[kb][r] : __builtin_amdgcn_exp2f(fmaf(s_cur[kb][r], scale_log2e, nm));
            const float a1 = (ABL & 2) ? s_cur[kb][r + 1] : __builtin_amdgcn_exp2f(fmaf(s_cur[kb][r + 1], scale_log2e, nm));
            psum += a0 + a1;
            const h2 a = __builtin_convertvector(f2{a0, a1}, h2);
            pf[u][e % 8] = a[0], pf[u][e % 8 + 1] = a[1];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      l_run += psum;
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ================= phase B_j: O^T += V_j^T P_j^T
    {
      h8 vf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) vf[i] = v_frag(j, i);
      if (!(ABL & 8)) {
#pragma unroll
        for (int idx = 0; idx < NPV; ++idx) {
          const int st = idx / (D / 32), b = idx % (D / 32);
          ot[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[idx % PD], pf[st], ot[b], 0, 0, 0);
          cln_mfma_keep(ot[b], vf[idx % PD], pf[st]);  // destination disjoint from the operands (common.h)
          if (idx + PD < NPV) vf[idx % PD] = v_frag(j, idx + PD);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    hgemm::wait_vmcnt<0>();  // own DMA pieces landed; behind the barrier nobody reads what the next phase overwrites
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  for (int j = 0; j < T; j += 2) {
    tile_step(j, s_a, s_b);
    tile_step(j + 1, s_b, s_a);
  }
  if (grp == 0) {
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
  for (int b = 0; b < D / 32; ++b) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][rq * 4 + e] * inv);
      *reinterpret_cast<h4*>(ob + l31 * G::OS + (b * 32 + rq * 8 + hi * 4) * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  constexpr int LPR = D / 8;
  half_t* og = O + head + (size_t)q_row0 * D;
#pragma unroll 4
  for (int it = 0; it < (32 * LPR) / 64; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / LPR, c = idx % LPR;
    const u4 v = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
    *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = v;
  }
}

template <int D, int BCB, int OPT, int ABL = 0>
int launch_pipe(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoSplit<D, 1, BCB>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_pipe_kernel<D, BCB, OPT, ABL>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_pipe_kernel<D, BCB, OPT, ABL>), dim3(n_qblk * B * H), dim3(G::NT), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
