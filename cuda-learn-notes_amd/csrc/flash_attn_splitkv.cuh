// FlashAttention-2 forward, SPLIT-KV rung: the structurally distinct first rung of the reference ladder
// (kernels/flash-attn/mma/basic/flash_attn_mma_split_kv.cu:31 -- "Split Q across MMA(Warps) and keep access KV for all
// MMA(Warps)" is the later rung; this one splits the KV tile across the warps, every warp holds the SAME query rows,
// and the row max / row sum are combined across warps through shared memory for every tile, :130-131).
//
// gfx950 form (not a translation of the m16n8k16 warp layout):
//  * a workgroup = 4 waves that ALL own the same 32 query rows; a KV tile is 128 keys and wave w owns keys
//    32w .. 32w+31 of it, so per tile each wave computes one 32x32 block of S^T = K Q^T (D/16 MFMAs on
//    v_mfma_f32_32x32x16_f16, swapped operands: a lane owns one query row) and one partial O^T += V_w^T P_w^T;
//  * K fragments come straight from global memory (a wave is the only reader of its 32 keys: nothing to share
//    through LDS); the wave's 32 V rows are staged in a WAVE-PRIVATE LDS image and fetched with the transposing
//    ds_read_b64_tr_b16 in the order the P registers already have;
//  * cross-wave softmax through LDS, once per tile: every wave publishes its 32 row maxima, ONE workgroup barrier,
//    every wave folds the four values into the shared running max (identical in all waves, so the partial
//    accumulators of the four waves stay on one scale and simply add up at the end). The exchange buffer is
//    double-buffered so a tile needs one barrier, not two;
//  * epilogue: the four partial O^T (fp32) and the four partial row sums are summed through LDS.
// This rung exists to make the ladder's first step measurable (it is the slow one: 32 query rows per workgroup
// re-read all of K and V); the split-Q kernels are the production path.
#pragma once
#include "common.h"

namespace fa2 {

template <int D>
struct GeoSplitKV {
  static constexpr int NW = 4, NT = 256, BR = 32, BC = 128;
  static constexpr int VPAD = ((D * 2) % 128 == 64) ? 0 : 64;  // row stride == 64 (mod 128) bytes: 4 rows tile the banks
  static constexpr int VS = D * 2 + VPAD;
  static constexpr int V_WAVE = 32 * VS;                // one wave's 32 V rows
  static constexpr int RED = 2 * NW * 32 * 4;           // double-buffered row maxima [2][wave][row]
  static constexpr int OS = (D + 1) * 4;                // fp32 partial-O row stride (+1: bank spread)
  static constexpr int OPART = NW * 32 * OS + NW * 32 * 4;  // partial O^T and partial row sums
  static constexpr int OFF_RED = NW * V_WAVE;
  static constexpr int OFF_O = OFF_RED + RED;
  static constexpr int LDS_BYTES = OFF_O + OPART;
  static_assert(D % 32 == 0 && D <= 128, "split-KV rung: head dims 32..128 (reference driver table)");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// PREFETCH (the `stages = 2` form; reference kStage, flash_attn_mma_split_kv.cu template parameter): the K fragments of tile j + 1 are
// loaded into a second set of registers while tile j is computed; `stages = 1` loads a tile, waits, then uses it.
template <int D, bool PREFETCH = false>
__global__ __launch_bounds__(256) void fa2_fwd_splitkv_kernel(const half_t* __restrict__ Q, const half_t* __restrict__ K,
                                                             const half_t* __restrict__ V, half_t* __restrict__ O,
                                                             int N, float scale_log2e) {
  using G = GeoSplitKV<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const size_t head = (size_t)blockIdx.y * N * D;
  const int q_row0 = blockIdx.x * G::BR;
  const half_t* Kh = K + head;
  const half_t* Vh = V + head;
  char* v_lds = smem + wave * G::V_WAVE;
  float* red = reinterpret_cast<float*>(smem + G::OFF_RED);

  // Q fragments (B operand of S^T = K Q^T): lane (q = l31) holds d = 16*ks + 8*hi .. +7; the same in all four waves
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
  float m_run = -INFINITY, l_run = 0.f;

  const int T = (N + G::BC - 1) / G::BC;
  // this wave's V rows of a tile -> registers (idx = lane + 64 u: row idx / (D/8), 16-byte chunk idx % (D/8)), its K fragments
  // straight from global: lane (key = l31) reads d = 16*ks + 8*hi .. +7
  auto load_v = [&](int key0, u4 (&vreg)[D / 16]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < D / 16; ++u) {
      const int idx = lane + u * 64, row = idx / (D / 8), ch = idx % (D / 8);
      vreg[u] = *reinterpret_cast<const u4*>(Vh + (size_t)(key0 + row) * D + ch * 8);
    }
  };
  auto load_k = [&](int key0, h8 (&kf)[D / 16]) __attribute__((always_inline)) {
    const half_t* kp = Kh + (size_t)(key0 + l31) * D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) kf[ks] = *reinterpret_cast<const h8*>(kp + ks * 16);
  };
  // PREFETCH keeps the NEXT tile's K fragments in flight in a second register set (they are what the first MFMA of a tile waits
  // for); the V rows are loaded at the top of their own tile and fly under its QK^T MFMAs in both forms. (Prefetching V as well costs a
  // third of the occupancy at D = 64 -- 144 -> 180 registers -- and measured 6 % slower than no prefetch.)
  h8 kf_cur[D / 16], kf_nxt[PREFETCH ? D / 16 : 1];
  u4 v_cur[D / 16];
  if constexpr (PREFETCH) {
    if (wave * 32 < N) load_k(wave * 32, kf_cur);
  }
  for (int j = 0; j < T; ++j) {
    const int key0 = j * G::BC + wave * 32;
    const bool active = key0 < N;  // wave-uniform; N % 32 == 0
    f16v s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    float mx = -INFINITY;
    if (active) {
      load_v(key0, v_cur);  // issued first: the loads fly under the QK^T MFMAs
      if constexpr (PREFETCH) {
        if (key0 + G::BC < N) load_k(key0 + G::BC, kf_nxt);
      } else {
        load_k(key0, kf_cur);
      }
      // ---- S^T block = K_w Q^T
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf_cur[ks], qf[ks], s, 0, 0, 0);
        cln_mfma_keep(s, kf_cur[ks], qf[ks]);  // destination disjoint from the operands (common.h)
      }
      // ---- this wave's V rows -> wave-private LDS image
#pragma unroll
      for (int u = 0; u < D / 16; ++u) {
        const int idx = lane + u * 64, row = idx / (D / 8), ch = idx % (D / 8);
        *reinterpret_cast<u4*>(v_lds + row * G::VS + ch * 16) = v_cur[u];
      }
      mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    }
    // ---- cross-wave row max through LDS (reference split_kv.cu:130-131): publish, one barrier, fold
    float* rj = red + (j & 1) * (G::NW * 32);
    if (hi == 0) rj[wave * 32 + l31] = mx;
    __syncthreads();
    float tmax = rj[l31];
#pragma unroll
    for (int w = 1; w < G::NW; ++w) tmax = fmaxf(tmax, rj[w * 32 + l31]);
    const float m_new = fmaxf(m_run, tmax * scale_log2e);  // tile 0 always has wave 0 active: m_new is finite
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    l_run *= alpha;
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[b][r] *= alpha;
    if (active) {
      float psum = 0.f;
      h8 pf[2];  // P^T fragments of the two 16-key k-steps, in register order
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m_new));
        psum += p;
        pf[r >> 3][r & 7] = (half_t)p;
      }
      l_run += psum;
      // ---- partial O^T += V_w^T P_w^T: keys this lane's P registers cover in k-step st: 16*st + 4*hi + {0..3, 8..11}
      const int i = lane & 15;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int kv_base = 16 * st + 4 * hi;
#pragma unroll
        for (int b = 0; b < D / 32; ++b) {
          const char* vp = v_lds + (kv_base + (i >> 2)) * G::VS + (b * 32 + ((lane >> 4) & 1) * 16 + (i & 3) * 4) * 2;
          const h8 vf = h8_cat(lds_read_tr16(vp), lds_read_tr16(vp + 8 * G::VS));
          ot[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[st], ot[b], 0, 0, 0);
          cln_mfma_keep(ot[b], vf, pf[st]);
        }
      }
    }
    if constexpr (PREFETCH) {
#pragma unroll
      for (int u = 0; u < D / 16; ++u) kf_cur[u] = kf_nxt[u];
    }
  }

  // ---- epilogue: O = (sum over waves of O^T_w) / (sum over waves and lane halves of l_w)
  float* opart = reinterpret_cast<float*>(smem + G::OFF_O);
  float* lpart = opart + G::NW * 32 * (D + 1);
  {
    const float l_w = l_run + __shfl_xor(l_run, 32, 64);
    if (hi == 0) lpart[wave * 32 + l31] = l_w;
    float* ow = opart + (wave * 32 + l31) * (D + 1);
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) ow[b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = ot[b][r];
  }
  __syncthreads();
  {
    constexpr int CPT = D / 8;  // columns per thread: 256 threads cover 32 rows x D columns
    const int row = tid >> 3, c0 = (tid & 7) * CPT;
    float l_tot = 0.f;
#pragma unroll
    for (int w = 0; w < G::NW; ++w) l_tot += lpart[w * 32 + row];
    const float inv = 1.0f / l_tot;
    half_t* og = O + head + (size_t)(q_row0 + row) * D + c0;
#pragma unroll
    for (int c = 0; c < CPT; c += 4) {
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < G::NW; ++w) acc += opart[(w * 32 + row) * (D + 1) + c0 + c + e];
        o[e] = (half_t)(acc * inv);
      }
      *reinterpret_cast<h4*>(og + c) = o;
    }
  }
}

template <int D, bool PREFETCH = false>
int launch_splitkv(const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t stream) {
  using G = GeoSplitKV<D>;
  if (N % 32 != 0) return CLN_ERR_UNSUPPORTED;
  static cln_lds_attr lds_attr;  // per device, thread-safe (common.h)
  if (G::LDS_BYTES > 48 * 1024 &&
      cln_ensure_lds(lds_attr, reinterpret_cast<const void*>(&fa2_fwd_splitkv_kernel<D, PREFETCH>), G::LDS_BYTES) != CLN_OK)
    return CLN_ERR_LAUNCH;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  CLN_LAUNCH((fa2_fwd_splitkv_kernel<D, PREFETCH>), dim3(N / G::BR, B * H), dim3(G::NT), G::LDS_BYTES, stream,
             (const half_t*)q, (const half_t*)k, (const half_t*)v, (half_t*)o, N, scale_log2e);
  return cln_check_launch();
}

}  // namespace fa2
