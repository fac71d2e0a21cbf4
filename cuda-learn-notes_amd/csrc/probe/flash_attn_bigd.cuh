// FlashAttention-2 forward for large head dims, register-resident form. PROBE ONLY since the d-split kernels
// (flash_attn_dsplit.cuh, flash_attn_dwide.cuh) took over D = 512 / 768 / 1024; kept, tested, as the measured
// alternative and as the home of the LDS-DMA swizzles those kernels reuse. Written for config C5
// ([1,32,4096,512]): the reference's
// "fine-grained QKV tiling" rungs (kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qkv.cu:70, :732;
// tiling_qk.cu:72) keep O(1) shared memory by streaming Q, K and V in 16-wide d slices and re-reading Q for
// every KV tile. On MI355X the register file is the big resource, so the roles are inverted:
//   * a workgroup = 4 waves, ONE wave per SIMD, each wave owns 32 query rows and the WHOLE 512-register
//     file: the full-width O^T accumulator (D/32 x 16 = 256 registers at D = 512) and the Q fragments
//     (D/16 x 4 = 128 registers) stay in registers for the whole kernel -- Q is read from HBM exactly
//     once, S is computed once (the previous large-D path recomputed S per output slice: 1.5x the MFMA work);
//   * KV tile = 32 keys; K and V tiles ([32][D] fp16 = 32 KiB each at D = 512) are double-buffered in LDS
//     (128 KiB) and filled by LDS-DMA (global_load_lds_dwordx4, no staging registers -- there are none to
//     spare), one barrier per tile, tile j+1 in flight while tile j is consumed;
//   * LDS images are lane-linear (DMA constraint); bank conflicts of the 1-KiB-strided rows are removed by
//     XOR-swizzling the 16-byte chunk index on the DMA *source* address and on the fragment read:
//     K (ds_read_b128, 32 rows at one chunk column):   chunk ^= row & 15
//     V (ds_read_b64_tr_b16, 4 rows x 2 column blocks): chunk ^= (row & 3) << 2
//   * softmax / P handling as in flash_attn_v2.cuh (lane-local rows, deferred max, packed RNE conversion).
#pragma once
#include "flash_attn_v2.cuh"
#include "hgemm_mfma.cuh"  // glds16_asm, lds_addr_of, wait_vmcnt

namespace fa2 {

// DV: width of the output / V column slice one workgroup produces (DV == D: everything in one workgroup; DV < D:
// the grid carries D/DV slices and every slice recomputes S -- the price for fitting O^T + Q in the register file).
template <int D, int DV>
struct GeoBig {
  static constexpr int BC = 32, NW = 4, BR = 128, NT = 256;
  static constexpr int ROW = D * 2;               // bytes per K row
  static constexpr int VROW = DV * 2;             // bytes per V-slice row
  static constexpr int KTILE = BC * ROW, VTILE = BC * VROW;
  static constexpr int STAGE = KTILE + VTILE;     // K + V
  static constexpr int RING = 2 * STAGE;
  static constexpr int OS = DV * 2 + 16;
  static constexpr int EPI = NW * 32 * OS;
  static constexpr int LDS_BYTES = RING > EPI ? RING : EPI;
  static constexpr int KPPW = KTILE / 1024 / NW;  // 1-KiB DMA pieces per wave per K image
  static constexpr int VPPW = VTILE / 1024 / NW;
  static constexpr int NS = D / DV;
  static_assert(D % 128 == 0 && D >= 256 && D <= 1024, "big-D kernel: D multiple of 128");
  static_assert(DV % 128 == 0 && D % DV == 0 && DV <= 512, "slice width");
  static_assert((KTILE / 1024) % NW == 0 && (VTILE / 1024) % NW == 0, "pieces split over the waves");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int D, int DV, int OPT>
__global__ __launch_bounds__(256, 1) void fa2_fwd_bigd_kernel(const half_t* __restrict__ Q,
                                                              const half_t* __restrict__ K,
                                                              const half_t* __restrict__ V, half_t* __restrict__ O,
                                                              int N, int n_qblk, int n_heads, float scale_log2e) {
  using G = GeoBig<D, DV>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  int head_i, qb, slice;
  {
    const int bid = blockIdx.x;
    int k;
    if ((OPT & OPT_XCD) && (n_heads & 7) == 0) {
      const int xcd = bid & 7;
      k = bid >> 3;
      slice = k % G::NS;
      k /= G::NS;
      head_i = (k / n_qblk) * 8 + xcd;
      qb = k - (k / n_qblk) * n_qblk;
    } else {
      slice = bid % G::NS;
      k = bid / G::NS;
      head_i = k / n_qblk;
      qb = k - head_i * n_qblk;
    }
  }
  const size_t head = (size_t)head_i * N * D;
  const int q_row0 = qb * G::BR + wave * 32;
  const int dv0 = slice * DV;
  const char* Kh = reinterpret_cast<const char*>(K + head);
  const char* Vh = reinterpret_cast<const char*>(V + head + dv0);

  // ---- DMA source offsets (bytes, relative to the tile origin), constant over the KV loop
  unsigned k_voff[G::KPPW], v_voff[G::VPPW];
#pragma unroll
  for (int i = 0; i < G::KPPW; ++i) {
    const int byte = (i * G::NW + wave) * 1024 + lane * 16;  // position in the linear LDS image
    const int row = byte / G::ROW, c = (byte % G::ROW) >> 4;
    k_voff[i] = (unsigned)(row * G::ROW + ((c ^ (row & 15)) << 4));
  }
#pragma unroll
  for (int i = 0; i < G::VPPW; ++i) {
    const int byte = (i * G::NW + wave) * 1024 + lane * 16;
    const int row = byte / G::VROW, c = (byte % G::VROW) >> 4;
    v_voff[i] = (unsigned)(row * G::ROW + ((c ^ ((row & 3) << 2)) << 4));  // source rows are D wide
  }
  const unsigned lds0 = hgemm::lds_addr_of(smem);
  auto dma_tile = [&](int j, int buf) {
    const char* ks = Kh + (size_t)j * G::KTILE;
    const char* vs = Vh + (size_t)j * G::KTILE;  // 32 source rows of D halves
    const unsigned kimg = lds0 + buf * G::STAGE, vimg = kimg + G::KTILE;
#pragma unroll
    for (int i = 0; i < G::KPPW; ++i) hgemm::glds16_asm(ks, k_voff[i], kimg + (unsigned)(i * G::NW + wave) * 1024u);
#pragma unroll
    for (int i = 0; i < G::VPPW; ++i) hgemm::glds16_asm(vs, v_voff[i], vimg + (unsigned)(i * G::NW + wave) * 1024u);
  };

  // ---- Q fragments: the whole head dim lives in registers
  h8 qf[D / 16];
  {
    const half_t* qp = Q + head + (size_t)(q_row0 + l31) * D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) qf[ks] = *reinterpret_cast<const h8*>(qp + ks * 16);
  }

  f16v ot[DV / 32];
#pragma unroll
  for (int b = 0; b < DV / 32; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[b][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const int T = N / G::BC;
  dma_tile(0, 0);
  // vmcnt(0) through the builtin, not inline asm: the compiler must SEE that the Q loads retired, or it guards every
  // first use of a Q fragment inside the KV loop with a counted vmcnt that also drains the tile prefetch just issued
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // ... and pin the fragments here: hipcc otherwise sinks the Q loads below the barrier and into the first KV iteration
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) asm volatile("" : "+v"(qf[ks]));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // Lane-constant LDS offsets. The swizzles only touch the low 4 bits of the 16-byte chunk index, so a read
  // address is (one of a few lane-constant registers) + (compile-time immediate): 8 registers for K, 4 for V.
  // (Computing the full XOR per fragment makes LICM hoist D/16 + D/32 address registers out of the KV loop
  // and the kernel spills -- the register file is full by design.)
  int koff[8];  // k-step ks reads chunk (2*ks + hi) ^ (row & 15):   low 4 bits from ks & 7, rest immediate
#pragma unroll
  for (int i = 0; i < 8; ++i) koff[i] = l31 * G::ROW + (((2 * i + hi) ^ (l31 & 15)) << 4);
  const int i16 = lane & 15;
  const int v_row = 4 * hi + (i16 >> 2);  // + 16*st (+ 8 for the second read): (row & 3) is the same for all
  int voff[4];  // output block b reads chunk (4*b + cc) ^ ((row & 3) << 2) = 4*(b ^ (row&3)) + cc
#pragma unroll
  for (int i = 0; i < 4; ++i)
    voff[i] = v_row * G::VROW + ((((i ^ (v_row & 3)) << 2) + (((lane >> 4) & 1) * 2) + ((i16 & 3) >> 1)) << 4) +
              ((i16 & 1) << 3);

  for (int j = 0; j < T; ++j) {
    const char* kb = smem + (j & 1) * G::STAGE;
    const char* vb = kb + G::KTILE;
    if (j + 1 < T) dma_tile(j + 1, (j + 1) & 1);  // its readers (tile j-1) finished before the last barrier

    // ---- S^T = K Q^T (32 keys x 32 queries) over D/16 k-steps
    f16v s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    auto k_frag = [&](int ks) { return *reinterpret_cast<const h8*>(kb + koff[ks & 7] + (ks >> 3) * 256); };
    auto v_frag = [&](int idx) {  // idx = st * (DV/32) + b
      const int st = idx / (DV / 32), b = idx % (DV / 32);
      // rows 16*st + v_row and + 8 ((row + 8) & 3 == row & 3: same swizzle); 4 output blocks = 256 bytes
      const char* vp = vb + voff[b & 3] + (16 * st) * G::VROW + (b >> 2) * 256;
      return h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::VROW));
    };
    constexpr int NPV = 2 * (DV / 32);
    constexpr int PD = 8;  // fragments kept in flight by the prefetching forms (one wave per SIMD: nobody else
                           // hides the ~128-cycle LDS latency)
    h8 vpre[PD];
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(1);
    if constexpr ((OPT & OPT_KPRE) != 0) {
      h8 kf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = k_frag(i);
      f16v s1;
      if constexpr ((OPT & OPT_STAGGER) != 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s1[r] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        if ((OPT & OPT_STAGGER) != 0 && (ks & 1)) s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks % PD], qf[ks], s1, 0, 0, 0);
        else s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks % PD], qf[ks], s, 0, 0, 0);
        cln_mfma_keep(s, kf[ks % PD], qf[ks]);  // destination disjoint from the operands (common.h)
        if ((OPT & OPT_STAGGER) != 0) cln_mfma_keep(s1, kf[ks % PD], qf[ks]);
        if (ks + PD < D / 16) kf[ks % PD] = k_frag(ks + PD);
        else if ((OPT & OPT_VPRE) != 0) vpre[ks + PD - D / 16] = v_frag(ks + PD - D / 16);  // V under the QK^T tail
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr ((OPT & OPT_STAGGER) != 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] += s1[r];
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        const h8 kf = k_frag(ks);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s, 0, 0, 0);
        cln_mfma_keep(s, kf, qf[ks]);  // destination disjoint from the operands (common.h)
        // fence the scheduler every 4 k-steps: without it all D/16 fragment reads are hoisted ahead of the MFMA
        // chain and the kernel spills (the register file is full by design)
        if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr ((OPT & OPT_VPRE) != 0) {
#pragma unroll
        for (int i = 0; i < PD; ++i) vpre[i] = v_frag(i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(0);

    // ---- online softmax
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
      // The accumulators live in AGPRs and VALU ops cannot read those: every element makes a round trip through a
      // VGPR. Left alone, the compiler batches all D/2 reads (256 VGPR temporaries at D = 512 -> spills in the
      // hot loop); the empty volatile asm statements serialise the round trips to 4 live temporaries.
#pragma unroll
      for (int b = 0; b < DV / 32; ++b)
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

    // ---- O^T += V^T P^T : 2 k-steps x DV/32 output blocks
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(1);
    if constexpr ((OPT & OPT_VPRE) != 0) {
#pragma unroll
      for (int idx = 0; idx < NPV; ++idx) {
        const int st = idx / (DV / 32), b = idx % (DV / 32);
        ot[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vpre[idx % PD], pf[st], ot[b], 0, 0, 0);
        cln_mfma_keep(ot[b], vpre[idx % PD], pf[st]);  // destination disjoint from the operands (common.h)
        if (idx + PD < NPV) vpre[idx % PD] = v_frag(idx + PD);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int idx = 0; idx < NPV; ++idx) {
        const int st = idx / (DV / 32), b = idx % (DV / 32);
        const h8 vf = v_frag(idx);
        ot[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[st], ot[b], 0, 0, 0);
        cln_mfma_keep(ot[b], vf, pf[st]);  // destination disjoint from the operands (common.h)
        if ((b & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr ((OPT & OPT_PRIO) != 0) __builtin_amdgcn_s_setprio(0);

    // tile j+1 landed (own pieces) + everyone is done reading buffer j&1
    hgemm::wait_vmcnt<0>();
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
  for (int b = 0; b < DV / 32; ++b) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][rq * 4 + e] * inv);
      *reinterpret_cast<h4*>(ob + l31 * G::OS + (b * 32 + rq * 8 + hi * 4) * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  constexpr int LPR = DV / 8;
  half_t* og = O + head + (size_t)q_row0 * D + dv0;
#pragma unroll 4
  for (int it = 0; it < (32 * LPR) / 64; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / LPR, c = idx % LPR;
    const u4 v = *reinterpret_cast<const u4*>(ob + row * G::OS + c * 16);
    *reinterpret_cast<u4*>(og + (size_t)row * D + c * 8) = v;
  }
}

template <int D, int DV, int OPT>
int launch_bigd(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoBig<D, DV>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_bigd_kernel<D, DV, OPT>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  const int n_qblk = N / G::BR;
  CLN_LAUNCH((fa2_fwd_bigd_kernel<D, DV, OPT>), dim3(n_qblk * B * H * G::NS), dim3(G::NT), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, n_qblk, B * H, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
