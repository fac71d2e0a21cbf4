// C-ABI entry points of the FlashAttention-2 forward library: one symbol per function exported by
// the reference's pybind module (kernels/flash-attn/pybind/flash_attn.cc:182-215).
//
//   int name(q, k, v, o, B, H, N, D, stages, stream)
//
// q,k,v,o: fp16 [B,H,N,D] contiguous; the *_swizzle_qkv variants of share_kv / share_qkv /
// tiling_qk take v TRANSPOSED, [B,H,D,N] (reference flash_attn_mma.py:377-378, :542-565).
// stages: 1 = load-then-compute per KV tile, 2 = next K/V tile prefetched under the MFMA phases
// (reference kStage template parameter, flash_attn_mma_share_qkv.cu:843-884).
#include "flash_attn.cuh"
#include "flash_attn_large_d.cuh"
#include "flash_attn_v2.cuh"

namespace {

template <bool VT>
int fa2_dispatch(const void* q, const void* k, const void* v, void* o, int B, int H, int N, int D, int stages,
                 int max_d, hipStream_t s) {
  if (!q || !k || !v || !o || B <= 0 || H <= 0 || N <= 0) return CLN_ERR_BAD_ARG;
  if (!cln_aligned16(q) || !cln_aligned16(k) || !cln_aligned16(v) || !cln_aligned16(o)) return CLN_ERR_BAD_ARG;
  if (D > max_d) return CLN_ERR_UNSUPPORTED;  // "headdim not support!"
  (void)stages;  // v2 always runs the double-buffered prefetch pipeline; `stages` 1 and 2 are the same kernel
  // Head dims 32..256: v2 kernel. Workgroup = 8 / 4 / 2 waves x 32 query rows by the divisibility of N
  // (reference: N % max(Br,Bc) == 0 with Br = 128 or 64, flash_attn_mma_share_qkv.cu:769, split_q.cu:754).
  // waves per workgroup: the largest of 8 / 4 / 2 (x 32 query rows) that N allows AND that still gives every one of
  // the 256 CUs a workgroup; small problems take the smaller workgroup (measured [2,8,2048,64]: 534 TF with
  // 4 waves x 256 workgroups vs 413 TF with 8 waves x 128 workgroups).
  const long long bh = (long long)B * H;
  int nw = 0;
  for (int cand : {8, 4, 2}) {
    if (N % (cand * 32) != 0) continue;
    nw = cand;
    if (bh * (N / (cand * 32)) >= 256) break;
  }
  if (nw == 0) return CLN_ERR_UNSUPPORTED;
#define FA_V2(DD, OPTT)                                                                            \
  case DD:                                                                                         \
    if (nw == 8) return fa2::launch_v2<DD, 8, VT, OPTT>(q, k, v, o, B, H, N, s);                   \
    if (nw == 4) return fa2::launch_v2<DD, 4, VT, OPTT>(q, k, v, o, B, H, N, s);                   \
    return fa2::launch_v2<DD, 2, VT, OPTT>(q, k, v, o, B, H, N, s);
  // Head dims 64 / 128 / 256 with enough 256-row workgroups to occupy most of the chip: two-group ping-pong kernel
  // (flash_attn_dsplit.cuh: 8 waves x 32 rows, K/V by LDS-DMA, the two 4-wave groups one phase apart).
  // D = 256: 1000-1180 TF vs 630-790 for v2 (which needs one wave per SIMD there); D = 128: 970-1100 vs 900-1045
  // (profiles/r01_fa_dsplit_d256_probe.log, r01_fa_dsplit_d128_probe.log, r01_fa_dsplit_d64_probe.log).
  if constexpr (!VT) {
    if (N % 256 == 0 && bh * (N / 256) >= 192) {
      if (D == 256) return fa2::launch_dsplit<256, 1, 1, fa2::OPT_DEFAULT | fa2::OPT_KPRE>(q, k, v, o, B, H, N, s);
      if (D == 128) return fa2::launch_dsplit<128, 1, 2, fa2::OPT_DEFAULT | fa2::OPT_KPRE>(q, k, v, o, B, H, N, s);
      // D = 64 (config C4): 128-key tiles, half of the exponentials moved into the QK^T phase (OPT_STAGGER):
      // 730-775 TF at [4,8,2048,64] vs 620-665 for v2, 950 vs 915-940 at [1,48,8192,64]
      if (D == 64) return fa2::launch_dsplit<64, 1, 4, 13 | fa2::OPT_STAGGER>(q, k, v, o, B, H, N, s);
    }
  }
  switch (D) {
    FA_V2(32, 13)
    FA_V2(64, 13)
    FA_V2(96, 15)
    FA_V2(128, 15)
    case 256:
      // v2 at D = 256 needs the whole register file (one wave per SIMD): 4 waves x 32 rows only
      if (N % 128 == 0) return fa2::launch_v2<256, 4, VT, 15>(q, k, v, o, B, H, N, s);
      return CLN_ERR_UNSUPPORTED;
    default:
      break;
  }
#undef FA_V2
  if constexpr (!VT) {
    if (D == 512 || D == 1024 || D == 768 || D == 320 || D == 384 || D == 640)
      return fa::launch_fa2_large_d(q, k, v, o, B, H, N, D, stages, s);
  }
  return CLN_ERR_UNSUPPORTED;
}

}  // namespace

#define CLN_FA(name, VT, MAXD)                                                                            \
  CLN_API int name(const void* q, const void* k, const void* v, void* o, int B, int H, int N, int D,     \
                   int stages, void* stream) {                                                            \
    return fa2_dispatch<VT>(q, k, v, o, B, H, N, D, stages, MAXD, (hipStream_t)stream);                   \
  }

// max head dim per function follows the reference driver table (flash_attn_mma.py:436-506)
CLN_FA(flash_attn_mma_stages_split_kv, false, 128)
CLN_FA(flash_attn_mma_stages_split_q, false, 128)
CLN_FA(flash_attn_mma_stages_split_q_shared_kv, false, 256)
CLN_FA(flash_attn_mma_stages_split_q_shared_qkv, false, 256)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qk, false, 1024)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qkv, false, 1024)
CLN_FA(flash_attn_mma_stages_split_q_shared_kv_acc_f32, false, 256)
CLN_FA(flash_attn_mma_stages_split_q_shared_qkv_acc_f32, false, 256)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qk_acc_f32, false, 1024)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32, false, 1024)
CLN_FA(flash_attn_mma_stages_split_q_shared_kv_swizzle_q, false, 256)
CLN_FA(flash_attn_mma_stages_split_q_shared_kv_swizzle_qk, false, 256)
CLN_FA(flash_attn_mma_stages_split_q_shared_kv_swizzle_qkv, true, 256)
CLN_FA(flash_attn_mma_stages_split_q_shared_qkv_swizzle_q, false, 256)
CLN_FA(flash_attn_mma_stages_split_q_shared_qkv_swizzle_qk, false, 256)
CLN_FA(flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv, true, 256)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qk_swizzle_q, false, 1024)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qk_swizzle_qk, false, 1024)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv, true, 256)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_q, false, 1024)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qk, false, 1024)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qkv, false, 1024)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_q, false, 1024)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qk, false, 1024)
CLN_FA(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qkv, false, 1024)
// BUILD_FLASH_ATTN_MMA_OTHERS set (flash_attn.cc:161-180) -- always built here
CLN_FA(flash_attn_mma_stages_split_q_shared_qkv_Os2g, false, 256)
CLN_FA(flash_attn_mma_stages_split_q_shared_kv_acc_f32_rr, false, 256)
CLN_FA(flash_attn_mma_stages_split_q_shared_qkv_acc_f32_rr, false, 256)
