// FlashAttention-2 forward for gfx950: O = softmax(Q K^T / sqrt(d)) V, non-causal, fp16 in/out,
// fp32 softmax statistics and fp32 accumulators. Q,K,V,O are [B,H,N,d] contiguous.
//
// Replaces the split-Q family of the reference:
//   kernels/flash-attn/mma/basic/flash_attn_mma_split_q.cu:52        (split-Q)
//   kernels/flash-attn/mma/basic/flash_attn_mma_share_kv.cu:66       (K/V aliasing one smem region)
//   kernels/flash-attn/mma/basic/flash_attn_mma_share_qkv.cu:66,:712 (Q resident in registers) -- config C4
//   kernels/flash-attn/mma/swizzle/*_swizzle_qkv.cu                   (V passed transposed [B,H,d,N])
//
// MI355X design (not a translation of the m16n8k16 warp tiling):
//  * a workgroup = 4 waves x 32 query rows (Br = 128), KV tile Bc = 64, v_mfma_f32_32x32x16_f16;
//  * QK^T is computed SWAPPED, S^T = K Q^T, so each lane owns ONE query row (column lane&31 of the
//    32x32 result) and holds 32 of its 64 scores: row max / row sum are 31 in-lane ops + one
//    cross-half exchange (lane ^ 32) instead of the reference's 4-lane shuffles per MMA row pair;
//  * P never leaves registers: P^T is the B operand of O^T = V^T P^T. The k-index of that MFMA is
//    a free permutation, so the V fragment is fetched in the kv order the lane's P registers are
//    already in (two ds_read_b64_tr_b16 transposing reads per fragment) -- no permute, no LDS trip;
//  * O^T accumulators have the lane's own query row as their column, so the online-softmax
//    rescale is a lane-local multiply;
//  * Q fragments live in registers for the whole kernel ("shared-QKV": Q needs no LDS at all);
//    K and V tiles go global -> VGPR -> LDS with issue-early / write-late staging, K padded by
//    16 B per row (conflict-free ds_read_b128 over 32 rows), V padded so 4 consecutive rows x 64 B
//    tile the 64 banks for the transposing read.
#pragma once
#include "common.h"

namespace fa {

// D  : head dim (QK^T contraction length)         DV : width of the output/V column slice this
// BC : KV tile rows (64, or 32 for very large D)        workgroup produces (== D except large D,
// VT : V passed transposed [B,H,D,N]                    where blockIdx.z walks D/DV slices)
template <int D, int DV, int BC, bool VT>
struct Geo {
  static constexpr int BR = 128, NT = 256;
  static constexpr int KS = D * 2 + 16;  // K row stride (bytes)
  // V image, row-major [kv][dv]: stride == 64 (mod 128) bytes so rows r..r+3 land on distinct
  // 64-byte bank slots.   V^T image [d][kv] (VT): BC*2-byte rows + 8.
  static constexpr int VPAD = ((DV * 2) % 128 == 64) ? 0 : 64;
  static constexpr int VS = VT ? (BC * 2 + 8) : (DV * 2 + VPAD);
  static constexpr int K_BYTES = BC * KS;
  static constexpr int V_BYTES = VT ? DV * VS : BC * VS;
  static constexpr int LDS_BYTES = K_BYTES + V_BYTES;
  static constexpr int KCH = BC * (D / 8) / NT;   // 16-byte chunks per thread per K tile
  static constexpr int VCH = BC * (DV / 8) / NT;  // ... per V tile
  static constexpr int NSUB = BC / 32;            // 32-kv sub-tiles per tile
  static_assert(D % 32 == 0 && DV % 32 == 0 && D % DV == 0, "head dim");
  static_assert(BC == 32 || BC == 64, "BC");
  static_assert(!VT || DV == D, "transposed V only for un-sliced heads");
  static_assert((BC * (D / 8)) % NT == 0 && (BC * (DV / 8)) % NT == 0, "tile must split over 256 threads");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int D, int DV, int BC, bool VT, bool PREFETCH>
__global__ __launch_bounds__(256) void fa2_fwd_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                      const half_t* __restrict__ V, half_t* __restrict__ O,
                                                      int N, float scale_log2e) {
  using G = Geo<D, DV, BC, VT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_lds = smem;
  char* v_lds = smem + G::K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const size_t head = (size_t)blockIdx.y * N * D;
  const int q_row = blockIdx.x * G::BR + wave * 32 + l31;
  const int dv0 = blockIdx.z * DV;
  const half_t* Kh = K + head;
  const half_t* Vh = V + head;

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q = l31) holds d = 16*ks + 8*hi .. +7
  h8 qf[D / 16];
  {
    const half_t* qp = Q + head + (size_t)q_row * D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) qf[ks] = *reinterpret_cast<const h8*>(qp + ks * 16);
  }

  // ---- staging helpers -------------------------------------------------------------------------
  u4 kreg[G::KCH], vreg[G::VCH];
  auto load_k = [&](int j) {
#pragma unroll
    for (int u = 0; u < G::KCH; ++u) {
      const int idx = tid + u * 256;
      const int row = idx / (D / 8), ch = idx % (D / 8);
      kreg[u] = *reinterpret_cast<const u4*>(Kh + (size_t)(j * BC + row) * D + ch * 8);
    }
  };
  auto write_k = [&]() {
#pragma unroll
    for (int u = 0; u < G::KCH; ++u) {
      const int idx = tid + u * 256;
      const int row = idx / (D / 8), ch = idx % (D / 8);
      *reinterpret_cast<u4*>(k_lds + row * G::KS + ch * 16) = kreg[u];
    }
  };
  auto load_v = [&](int j) {
#pragma unroll
    for (int u = 0; u < G::VCH; ++u) {
      const int idx = tid + u * 256;
      if constexpr (VT) {  // V^T [d][N]: tile = D rows x BC kv
        const int row = idx / (BC / 8), ch = idx % (BC / 8);
        vreg[u] = *reinterpret_cast<const u4*>(Vh + (size_t)row * N + j * BC + ch * 8);
      } else {
        const int row = idx / (DV / 8), ch = idx % (DV / 8);
        vreg[u] = *reinterpret_cast<const u4*>(Vh + (size_t)(j * BC + row) * D + dv0 + ch * 8);
      }
    }
  };
  auto write_v = [&]() {
#pragma unroll
    for (int u = 0; u < G::VCH; ++u) {
      const int idx = tid + u * 256;
      if constexpr (VT) {
        const int row = idx / (BC / 8), ch = idx % (BC / 8);
        *reinterpret_cast<u4*>(v_lds + row * G::VS + ch * 16) = vreg[u];
      } else {
        const int row = idx / (DV / 8), ch = idx % (DV / 8);
        *reinterpret_cast<u4*>(v_lds + row * G::VS + ch * 16) = vreg[u];
      }
    }
  };

  // ---- accumulators ----------------------------------------------------------------------------
  f16v ot[DV / 32];  // O^T[d-block]: column = own q row, rows = d
#pragma unroll
  for (int b = 0; b < DV / 32; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[b][r] = 0.f;
  float m_run = -INFINITY;  // running max in the scaled (log2) domain
  float l_run = 0.f;        // per-lane partial row sum (this lane's half of each kv tile)

  const int T = N / BC;
  if constexpr (PREFETCH) {
    load_k(0);
    write_k();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // Q + K_0 complete before the loop (see flash_attn_v2.cuh)
    load_v(0);
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }

  for (int j = 0; j < T; ++j) {
    if constexpr (PREFETCH) {
      __syncthreads();  // K_j visible; every wave is done with V_{j-1}
      write_v();        // V_j (its loads were issued one PV phase ago)
      if (j + 1 < T) load_k(j + 1);
    } else {
      __syncthreads();
      load_k(j);
      write_k();
      load_v(j);
      write_v();
      __syncthreads();
    }

    // ---- S^T = K Q^T : NSUB 32-kv sub-tiles -----------------------------------------------------
    f16v s[G::NSUB];
#pragma unroll
    for (int t = 0; t < G::NSUB; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
      const char* kp = k_lds + (t * 32 + l31) * G::KS + hi * 16;
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        const h8 kf = *reinterpret_cast<const h8*>(kp + ks * 32);
        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[t], 0, 0, 0);
        cln_mfma_keep(s[t], kf, qf[ks]);  // destination disjoint from the operands (common.h)
      }
    }

    // ---- online softmax (lane-local row) --------------------------------------------------------
    float mx = s[0][0];
#pragma unroll
    for (int t = 0; t < G::NSUB; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * scale_log2e);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    h8 pf[2 * G::NSUB];  // P^T fragments for the 16-kv k-steps, in register order
#pragma unroll
    for (int t = 0; t < G::NSUB; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], scale_log2e, -m_new));
        psum += p;
        pf[t * 2 + (r >> 3)][r & 7] = (half_t)p;
      }
    l_run = fmaf(l_run, alpha, psum);
#pragma unroll
    for (int b = 0; b < DV / 32; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[b][r] *= alpha;

    if constexpr (PREFETCH) {
      __syncthreads();  // V_j visible; every wave is done reading K_j
      if (j + 1 < T) {
        write_k();      // K_{j+1}
        load_v(j + 1);  // lands during P V
      }
    }

    // ---- O^T += V^T P^T -------------------------------------------------------------------------
#pragma unroll
    for (int st = 0; st < 2 * G::NSUB; ++st) {
      // kv rows this lane's P registers cover in k-step st: base + {0..3} and base + 8 + {0..3}
      const int kv_base = 32 * (st >> 1) + 16 * (st & 1) + 4 * hi;
#pragma unroll
      for (int b = 0; b < DV / 32; ++b) {
        h8 vf;
        if constexpr (VT) {
          const char* vp = v_lds + (b * 32 + l31) * G::VS + kv_base * 2;
          vf = h8_cat(*reinterpret_cast<const h4*>(vp), *reinterpret_cast<const h4*>(vp + 16));
        } else {
          const int i = lane & 15;
          const char* vp =
              v_lds + (kv_base + (i >> 2)) * G::VS + (b * 32 + ((lane >> 4) & 1) * 16 + (i & 3) * 4) * 2;
          vf = h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::VS));
        }
        ot[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[st], ot[b], 0, 0, 0);
        cln_mfma_keep(ot[b], vf, pf[st]);
      }
    }
  }

  // ---- epilogue: O = O^T / l --------------------------------------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  half_t* op = O + head + (size_t)q_row * D + dv0;
#pragma unroll
  for (int b = 0; b < DV / 32; ++b)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[b][rq * 4 + e] * inv);
      *reinterpret_cast<h4*>(op + b * 32 + rq * 8 + hi * 4) = o;
    }
}

template <int D, int DV, int BC, bool VT, bool PREFETCH>
int launch_fa2(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = Geo<D, DV, BC, VT>;
  if (N % G::BR != 0) return CLN_ERR_UNSUPPORTED;  // reference asserts N % max(Br,Bc) == 0 (share_qkv.cu:769)
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (G::LDS_BYTES > 48 * 1024 && cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_kernel<D, DV, BC, VT, PREFETCH>), G::LDS_BYTES) != CLN_OK) return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  CLN_LAUNCH((fa2_fwd_kernel<D, DV, BC, VT, PREFETCH>), dim3(N / G::BR, B * H, D / DV), dim3(256),
                     G::LDS_BYTES, stream, (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N,
                     scale_log2e);
  return cln_check_launch();
}

}  // namespace fa
