// Head dims above 256 ("fine-grained tiling" rungs of the reference:
// kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qk.cu:72, flash_attn_mma_tiling_qkv.cu:70 --
// config C5 is D = 512). Round-1 implementation: the same split-Q kernel with the whole K row in
// LDS (160 KiB LDS makes the reference's 16-wide d-slicing of K unnecessary up to D = 1024) and
// the OUTPUT head dim sliced across blockIdx.z (each workgroup recomputes S for its slice).
#pragma once
#include "flash_attn.cuh"
#include "flash_attn_bigd.cuh"
#include "flash_attn_dsplit.cuh"
#include "flash_attn_dwide.cuh"

namespace fa {
inline int launch_fa2_large_d(const void* q, const void* k, const void* v, void* o, int B, int H, int N, int D,
                              int stages, hipStream_t s) {
  (void)stages;
  switch (D) {
    case 320: return launch_fa2<320, 160, 64, false, false>(q, k, v, o, B, H, N, s);
    case 384: return launch_fa2<384, 192, 64, false, false>(q, k, v, o, B, H, N, s);
    // D = 512 (config C5): pairs of waves split the head dim, two 4-wave groups one phase apart
    // (flash_attn_dsplit.cuh): 990-1000 TF at [1,32,4096,512] vs 487 for the register-resident O-slice kernel
    // (flash_attn_bigd.cuh, still used for D = 768) and 411 for the v1 path (profiles/r01_fa_dsplit_probe.log)
    case 512: return fa2::launch_dsplit<512, 2, 1, fa2::OPT_DEFAULT>(q, k, v, o, B, H, N, s);
    case 640: return launch_fa2<640, 320, 32, false, false>(q, k, v, o, B, H, N, s);
    // D = 768 / 1024: three / four waves split the head dim of a 32-row group (flash_attn_dwide.cuh): 675-690 TF
    // at [1,16,4096,768] (big-D kernel 223-295), 706-776 TF at D = 1024 (v1 path 100-131, below torch SDPA)
    case 768: return fa2::launch_dwide<768, fa2::OPT_DEFAULT>(q, k, v, o, B, H, N, s);
    case 1024: return fa2::launch_dwide<1024, fa2::OPT_DEFAULT>(q, k, v, o, B, H, N, s);
    default: return CLN_ERR_UNSUPPORTED;
  }
}
}  // namespace fa
