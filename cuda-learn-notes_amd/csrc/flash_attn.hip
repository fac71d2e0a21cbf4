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

namespace {

template <bool VT>
int fa2_dispatch(const void* q, const void* k, const void* v, void* o, int B, int H, int N, int D, int stages,
                 int max_d, hipStream_t s) {
  if (!q || !k || !v || !o || B <= 0 || H <= 0 || N <= 0) return CLN_ERR_BAD_ARG;
  if (!cln_aligned16(q) || !cln_aligned16(k) || !cln_aligned16(v) || !cln_aligned16(o)) return CLN_ERR_BAD_ARG;
  if (D > max_d) return CLN_ERR_UNSUPPORTED;  // "headdim not support!"
  const bool pf = stages >= 2;
#define FA_CASE(DD)                                                                  \
  case DD:                                                                           \
    return pf ? fa::launch_fa2<DD, DD, 64, VT, true>(q, k, v, o, B, H, N, s)         \
              : fa::launch_fa2<DD, DD, 64, VT, false>(q, k, v, o, B, H, N, s);
  switch (D) {
    FA_CASE(32)
    FA_CASE(64)
    FA_CASE(96)
    FA_CASE(128)
    FA_CASE(256)
    default:
      break;
  }
#undef FA_CASE
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
