"""Exported-name manifest: every function the reference's extension modules export, the C-ABI
signature class it has here, and the gfx950 kernel that implements it.

Generated from / checked against the reference PYBIND11_MODULE blocks:
  kernels/hgemm/pybind/hgemm.cc:58-107, kernels/flash-attn/pybind/flash_attn.cc:182-215,
  kernels/elementwise/elementwise.cu:170-177, kernels/reduce/block_all_reduce.cu:792-813,
  kernels/softmax/softmax.cu:873-885, kernels/layer-norm/layer_norm.cu:804-814,
  kernels/rms-norm/rms_norm.cu:803-813, kernels/rope/rope.cu:116-120.

`impl` names the distinct HIP kernel (template + parameters); entries sharing an `impl` are aliases:
reference variants that differ only by an NVIDIA mechanism (ldmatrix.x2 vs .x4, register trimming,
stmatrix, WMMA-vs-MMA API, CuTe, smem swizzle-vs-pad) map onto the same CDNA4 kernel.
"""
from collections import namedtuple

Entry = namedtuple("Entry", "name sig lib impl")

# signature classes (C side):
#   G3   int f(a, b, c, M, N, K, stream)
#   G6   int f(a, b, c, M, N, K, stages, swizzle, swizzle_stride, stream)
#   H0   int f(void)
#   FA   int f(q, k, v, o, B, H, N, D, stages, stream)
#   P3   int f(a, b, c, n, stream)
#   R1   int f(a, y, n, stream)                 (python: Tensor f(x))
#   SG   int f(x, y, total_ws, n, stream)       (python: f(x, y))
#   XY   int f(x, y, S, H, stream)
#   LN   int f(x, y, g, b, N, K, stream)
#   RN   int f(x, y, g, N, K, stream)
#   RP   int f(x, out, seq_len, hidden, ref_quirk, stream)

_E = []


def _add(lib, sig, impl, *names):
    for n in names:
        _E.append(Entry(n, sig, lib, impl))


# ---------------------------------------------------------------- hgemm (38)
_add("hgemm", "G3", "valu_naive(1 elem/thread)", "hgemm_naive_f16")
_add("hgemm", "G3", "valu_sliced_k(32x32x32 LDS)", "hgemm_sliced_k_f16")
_add("hgemm", "G3", "valu_tile<BK=8,TM=8,single-buffer> v_dot2_f32_f16",
     "hgemm_t_8x8_sliced_k_f16x4", "hgemm_t_8x8_sliced_k_f16x4_pack", "hgemm_t_8x8_sliced_k_f16x4_bcf",
     "hgemm_t_8x8_sliced_k_f16x4_pack_bcf", "hgemm_t_8x8_sliced_k_f16x8_pack_bcf")
_add("hgemm", "G3", "valu_tile<BK=8,TM=8,dbuf>", "hgemm_t_8x8_sliced_k_f16x8_pack_bcf_dbuf")
_add("hgemm", "G3", "valu_tile<BK=16,TM=8,dbuf>", "hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf")
_add("hgemm", "G3", "valu_tile<BK=16,TM=8,dbuf,issue-early/write-late>",
     "hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf_async")
_add("hgemm", "G3", "valu_tile<BK=32,TM=8,dbuf>", "hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf")
_add("hgemm", "G3", "valu_tile<BK=32,TM=8,dbuf,issue-early/write-late>",
     "hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf_async")
_add("hgemm", "G3", "valu_tile<BK=32,TM=16,dbuf>", "hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf")
_add("hgemm", "G3", "valu_tile<BK=32,TM=16,dbuf,issue-early/write-late>",
     "hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf_async")
_add("hgemm_vendor", "H0", "rocblas_create_handle", "init_cublas_handle")
_add("hgemm_vendor", "H0", "rocblas_destroy_handle", "destroy_cublas_handle")
_add("hgemm_vendor", "G3", "rocblas_gemm_ex f16/f32-acc NN", "hgemm_cublas_tensor_op_nn")
_add("hgemm_vendor", "G3", "rocblas_gemm_ex f16/f32-acc TN", "hgemm_cublas_tensor_op_tn")
# second vendor comparison row (NOT a reference name, hence the cln_ prefix): hipBLASLt, csrc/hgemm_vendor_lt.hip
_add("hgemm_vendor_lt", "G3", "hipblasLtMatmul f16/f32-acc NN, heuristic's top algorithm", "cln_hgemm_hipblaslt_nn")
_add("hgemm_vendor_lt", "G3", "hipblasLtMatmul f16/f32-acc TN, heuristic's top algorithm", "cln_hgemm_hipblaslt_tn")
# attention comparison row (NOT a reference name): AMD's ck_tile FMHA forward kernels -- what FlashAttention-2-ROCm / aiter
# dispatch to -- instantiated from /opt/rocm/include/ck_tile; the `stages` slot of the signature carries the variant
# (0 = async pipeline, D = 64 / 128; 3 = the gfx950 "v3" kernel, D = 128). csrc/fa2_vendor_ck.hip
_add("fa2_vendor_ck", "FA", "ck_tile::FmhaFwdKernel<BlockFmhaPipelineQRKSVSAsync> (D = 64 / 128) | ck_tile::FmhaFwdV3Kernel (D = 128, variant 3)",
     "cln_fa2_ck_tile_fwd")
_add("hgemm", "G3", "mfma_naive<NN> 1 wave/16x16 tile, mfma_16x16x16",
     "hgemm_wmma_m16n16k16_naive", "hgemm_mma_m16n8k16_naive")
_add("hgemm", "G3", "mfma_1stage<64x128x32,2 waves,NN>", "hgemm_wmma_m16n16k16_mma4x2")
_add("hgemm", "G3", "mfma_1stage<128x128x32,4 waves,NN>; below 256 tiles of 128x128 (e.g. config C2, 1024^3): mfma_1stage<64x64x64,4 waves,NN>",
     "hgemm_wmma_m16n16k16_mma4x2_warp2x4", "hgemm_mma_m16n8k16_mma2x4_warp4x4")
_add("hgemm", "G3", "mfma_ring<128x128,BK=64|32,stages=2,NN>", "hgemm_wmma_m16n16k16_mma4x2_warp2x4_dbuf_async")
_add("hgemm", "G3", "mfma_ring<128x128,BK=32,stages=2,NN>", "hgemm_wmma_m32n8k16_mma2x4_warp2x4_dbuf_async")
_add("hgemm", "G6", "mfma_ring<128x128,BK by stages,NN>",
     "hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages", "hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem",
     "hgemm_mma_m16n8k16_mma2x4_warp4x4_stages", "hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem")
_add("hgemm", "G6", "hgemm_w4<256x128> at stages=2 (K % 64 == 0, K >= 384 / 448 for an even / odd K / 64) | mfma_ring<256x128,8 waves,NN>", "hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem")
_add("hgemm", "G6", "hgemm_w4<256x256> at stages=2 (K % 64 == 0, K >= 384 / 448 for an even / odd K / 64) | hgemm_w4s<256x256, ring of `stages` 32-deep slots> at stages 3 / 4 / 5 | mfma_ring<256x256,8 waves,NN>", "hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem")
_add("hgemm", "G6", "best<NN>: tile shape by estimated CU utilisation x kernel efficiency (csrc/hgemm.hip best_plan): hgemm_w4<256x256x64, one wave per SIMD> (stages 2) / hgemm_w4s<256x256, ring of `stages` 32-deep K slots, one wave per SIMD> (stages 3 / 4 / 5; bit-identical) -- hgemm_pp<256x256x64> / hgemm_pp32<4x32 ring> (stages 4) when K is < 384 or has an odd number < 7 of 64-wide tiles | hgemm_pp<192x256x64> | mfma_ring<128x256> | <64x128> | <64x64> (small problems) | <128x128> | split-K over hgemm_w4, reduced in the same launch by the last-arriving workgroup of a tile at 2 splits and by hgemm_splitk_reduce above (K >= 4096 and M N <= 2048^2: few tiles, long K; fp32 partials in a per-stream workspace: library-owned, bounded and freeable, or the caller's -- cln_hgemm_set_workspace) | tail split (a few 256x256 tiles past whole rounds of 256: the last tile rows as split-K) (see DISPATCH_EXAMPLES)",
     "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_x4",
     "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_rr",
     "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_swizzle")
_add("hgemm", "G6", "mfma_ring<128x128,TN>", "hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem_tn")
_add("hgemm", "G6", "best<TN>: hgemm_w4 / hgemm_w4s / hgemm_pp / hgemm_pp32 / mfma_ring / split-K as for NN (see DISPATCH_EXAMPLES)", "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4")
_add("hgemm", "G6", "hgemm_w4<128x256> at stages=2 (K % 64 == 0, K >= 384 / 448 for an even / odd K / 64) | mfma_ring<128x256,8 waves,TN>", "hgemm_mma_stages_block_swizzle_tn_cute")

# ---------------------------------------------------------------- flash-attn (28 + 3)
_FA_PLAIN = [
    "flash_attn_mma_stages_split_kv", "flash_attn_mma_stages_split_q",
    "flash_attn_mma_stages_split_q_shared_kv", "flash_attn_mma_stages_split_q_shared_qkv",
    "flash_attn_mma_stages_split_q_tiling_qk", "flash_attn_mma_stages_split_q_tiling_qkv",
    "flash_attn_mma_stages_split_q_shared_kv_acc_f32", "flash_attn_mma_stages_split_q_shared_qkv_acc_f32",
    "flash_attn_mma_stages_split_q_tiling_qk_acc_f32", "flash_attn_mma_stages_split_q_tiling_qkv_acc_f32",
    "flash_attn_mma_stages_split_q_shared_kv_swizzle_q", "flash_attn_mma_stages_split_q_shared_kv_swizzle_qk",
    "flash_attn_mma_stages_split_q_shared_qkv_swizzle_q", "flash_attn_mma_stages_split_q_shared_qkv_swizzle_qk",
    "flash_attn_mma_stages_split_q_tiling_qk_swizzle_q", "flash_attn_mma_stages_split_q_tiling_qk_swizzle_qk",
    "flash_attn_mma_stages_split_q_tiling_qkv_swizzle_q", "flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qk",
    "flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qkv",
    "flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_q",
    "flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qk",
    "flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qkv",
    "flash_attn_mma_stages_split_q_shared_qkv_Os2g", "flash_attn_mma_stages_split_q_shared_kv_acc_f32_rr",
    "flash_attn_mma_stages_split_q_shared_qkv_acc_f32_rr",
]
_FA_VT = [
    "flash_attn_mma_stages_split_q_shared_kv_swizzle_qkv", "flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv",
    "flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv",
]
_FA_SPLIT_Q_IMPL = ("planner fa2_plan(shape, stages) [csrc/flash_attn.hip]: stages=1 -> the stage-2 kernel of the shape in its single-stage form (each tile requested in one burst and waited for where it is requested; bit-identical to stages=2); "
                    "stages=2 -> fa2_fwd_m16x<D=128> / fa2_fwd_m16<D=256> (N a multiple of 256) | fa2_fwd_m16x<D=64> (> 128 workgroups of 256 rows) | fa2_fwd_pair2<D=512, and 320 / 384 on the D = 512 LDS geometry with every loop over the real head dim> (pairs of waves: rows split for QK^T and the softmax, d for PV) | fa2_fwd_m16x64r<D=64> (>=256 workgroups of 512 rows) | fa2_fwd_v2<D<=256, 4 waves when N allows, 8 at D=32|96 on large grids, 2 when N is an odd multiple of 64> | "
                    "fa2_fwd_dw4<640 | 768 | 1024: one wave per SIMD, O^T in AGPRs, softmax once per row>; mfma_32x32x16 / 16x16x32, f32 acc; "
                    "the *_acc_f32 names at D <= 128: the same kernels with the scores scaled in fp32 instead of the fp16 pre-scaled Q "
                    "(see DISPATCH_EXAMPLES)")
_add("flash_attn", "FA", "fa2_fwd_splitkv<D<=128>: 4 waves share 32 query rows, KV tile split over the waves, cross-wave "
     "row max through LDS (the structurally distinct split-KV rung)", "flash_attn_mma_stages_split_kv")
_add("flash_attn", "FA", _FA_SPLIT_Q_IMPL, *[n for n in _FA_PLAIN if n != "flash_attn_mma_stages_split_kv"])
_add("flash_attn", "FA", "planner as above with V transposed [B,H,D,N]: fa2_fwd_m16x<D=64|128,V^T> / fa2_fwd_m16x64r<V^T> | fa2_fwd_v2<V^T> (stages=1: their single-stage forms)",
     *_FA_VT)

# max head dim per function (reference flash_attn_mma.py:436-506; C side enforces the same)
FA_MAX_HEADDIM = {n: 256 for n in _FA_PLAIN + _FA_VT}
for _n in _FA_PLAIN:
    if "tiling" in _n:
        FA_MAX_HEADDIM[_n] = 1024
FA_MAX_HEADDIM["flash_attn_mma_stages_split_kv"] = 128
FA_MAX_HEADDIM["flash_attn_mma_stages_split_q"] = 128
FA_V_TRANSPOSED = set(_FA_VT)

# ---------------------------------------------------------------- bandwidth kernels
_add("elementwise", "P3", "add_vec<float,4B>", "elementwise_add_f32")
_add("elementwise", "P3", "add_vec<float4,16B>", "elementwise_add_f32x4")
_add("elementwise", "P3", "add_vec<half,2B>", "elementwise_add_f16")
_add("elementwise", "P3", "add_vec<half2,4B>", "elementwise_add_f16x2")
_add("elementwise", "P3", "add_vec<half2,4B> with four times the packs per lane (eight halves as four coalesced 4-byte accesses)", "elementwise_add_f16x8")
_add("elementwise", "P3", "add_vec<half8,16B>", "elementwise_add_f16x8_pack")

for _n, _impl in [
    ("f32_f32", "f32,1"), ("f32x4_f32", "f32,4"), ("f16_f16", "f16 acc f16,1"), ("f16_f32", "f16 acc f32,1"),
    ("f16x2_f16", "f16 acc f16,2"), ("f16x2_f32", "f16 acc f32,2"), ("f16x8_pack_f16", "f16 acc f16,8"),
    ("f16x8_pack_f32", "f16 acc f32,8"), ("bf16_bf16", "bf16 acc bf16,1"), ("bf16_f32", "bf16 acc f32,1"),
    ("bf16x2_bf16", "bf16 acc bf16,2"), ("bf16x2_f32", "bf16 acc f32,2"), ("bf16x8_pack_bf16", "bf16 acc bf16,8"),
    ("bf16x8_pack_f32", "bf16 acc f32,8"), ("fp8_e4m3_f16", "e4m3 acc f16,1"),
    ("fp8_e4m3x16_pack_f16", "e4m3 acc f16,16"), ("fp8_e5m2_f16", "e5m2 acc f16,1"),
    ("fp8_e5m2x16_pack_f16", "e5m2 acc f16,16"), ("i8_i32", "i8 acc i32,1"), ("i8x16_pack_i32", "i8 acc i32,16"),
]:
    _add("reduce", "R1", "reduce_sum<%s>" % _impl, "block_all_reduce_sum_" + _n)

_add("softmax", "SG", "exp_sum<1> + exp_div<1>", "softmax_f32")
_add("softmax", "SG", "exp_sum<4> + exp_div<4>", "softmax_f32x4")
for _n, _impl in [
    ("softmax_f32_per_token", "float,1,unsafe"), ("softmax_f32x4_per_token", "float,4,unsafe"),
    ("safe_softmax_f32_per_token", "float,1,safe"), ("safe_softmax_f32x4_per_token", "float,4,safe"),
    ("safe_softmax_f16_f32_per_token", "half,1,safe"), ("safe_softmax_f16x2_f32_per_token", "half,2,safe"),
    ("safe_softmax_f16x8_pack_f32_per_token", "half,8,safe"),
    ("online_safe_softmax_f32_per_token", "float,1,online"),
    ("online_safe_softmax_f32x4_pack_per_token", "float,4,online"),
]:
    _add("softmax", "XY", "softmax_row<%s>" % _impl, _n)

for _n, _impl in [
    ("layer_norm_f32", "float,1"), ("layer_norm_f32x4", "float,4"), ("layer_norm_f16_f16", "half,1"),
    ("layer_norm_f16_f32", "half,1"), ("layer_norm_f16x2_f16", "half,2"), ("layer_norm_f16x8_f16", "half,8"),
    ("layer_norm_f16x8_pack_f16", "half,8"), ("layer_norm_f16x8_pack_f32", "half,8"),
]:
    _add("layer_norm", "LN", "layer_norm_row<%s> fp32 stats" % _impl, _n)
for _n, _impl in [
    ("rms_norm_f32", "float,1"), ("rms_norm_f32x4", "float,4"), ("rms_norm_f16_f16", "half,1"),
    ("rms_norm_f16x2_f16", "half,2"), ("rms_norm_f16x8_f16", "half,8"), ("rms_norm_f16x8_pack_f16", "half,8"),
    ("rms_norm_f16x8_f32", "half,8"), ("rms_norm_f16x8_pack_f32", "half,8"), ("rms_norm_f16_f32", "half,1"),
]:
    _add("rms_norm", "RN", "rms_norm_row<%s> fp32 stats" % _impl, _n)

_add("rope", "RP", "rope<1 pair/thread, 8B>", "rope_f32", "rope_f32_v2")
_add("rope", "RP", "rope<2 pairs/thread, 16B>", "rope_f32x4_pack")

# ---------------------------------------------------------------- bit-exact indexing kernels
#   HI   int f(a, y, n, nbins, stream)                     (python: Tensor f(a))
#   EM   int f(idx, weight, out, n, emb_size, vocab, stream)  (python: f(a, weight, o))
_add("histogram", "HI", "histogram<4 B/lane> LDS-privatised counters, flush non-zero bins", "histogram_i32")
_add("histogram", "HI", "histogram<16 B/lane> LDS-privatised counters, flush non-zero bins", "histogram_i32x4")
_add("embedding", "EM", "row gather<float,1>", "embedding_f32")
_add("embedding", "EM", "row gather<float,4 scalar accesses>", "embedding_f32x4")
_add("embedding", "EM", "row gather<float,16 B pack>", "embedding_f32x4_pack")
_add("embedding", "EM", "row gather<half,1>", "embedding_f16")
_add("embedding", "EM", "row gather<half,8 scalar accesses>", "embedding_f16x8")
_add("embedding", "EM", "row gather<half,16 B pack>", "embedding_f16x8_pack")

# ---------------------------------------------------------------- activation family (7 x 6)
#   UN   int f(x, y, n, stream)                            (python: f(x, y))
ACTIVATIONS = ("relu", "sigmoid", "gelu", "swish", "elu", "hardswish", "hardshrink")
for _op in ACTIVATIONS:
    for _r, _impl in (("f32", "float,4 B"), ("f32x4", "float,16 B"), ("f16", "half,2 B"), ("f16x2", "half,4 B"),
                      ("f16x8", "half,4 x 4 B"), ("f16x8_pack", "half,16 B")):
        _add("activation", "UN", "unary<%s,%s> fp32 math" % (_op, _impl), "%s_%s" % (_op, _r))

# ---------------------------------------------------------------- dot / gemv / transpose
#   D2   int f(a, b, y, n, stream)              (python: Tensor f(a, b))
#   GV   int f(a, x, y, M, K, stream)           (python: f(a, x, y))
#   TR   int f(x, y, row, col, stream)          (python: f(x, y))
for _n, _impl in (("dot_prod_f32_f32", "float,4 B"), ("dot_prod_f32x4_f32", "float,16 B"), ("dot_prod_f16_f32", "half,2 B"),
                  ("dot_prod_f16x2_f32", "half,4 B"), ("dot_prod_f16x8_pack_f32", "half,16 B")):
    _add("dot_product", "D2", "dot<%s> fp32 acc, 1 workgroup/CU" % _impl, _n)
for _n, _impl in (("sgemv_k32_f32", "float,1,32 lanes/row"), ("sgemv_k128_f32x4", "float,4,32 lanes/row"),
                  ("sgemv_k16_f32", "float,1,16 lanes/row")):
    _add("sgemv", "GV", "gemv<%s> fp32 acc" % _impl, _n)
for _n, _impl in (("hgemv_k32_f16", "half,1,32 lanes/row"), ("hgemv_k128_f16x4", "half,4,32 lanes/row"),
                  ("hgemv_k16_f16", "half,1,16 lanes/row")):
    _add("hgemv", "GV", "gemv<%s> fp32 acc" % _impl, _n)
_add("mat_transpose", "TR", "transpose read-coalesced<1>", "mat_transpose_f32_col2row", "mat_transpose_f32_col2row2d")
_add("mat_transpose", "TR", "transpose read-coalesced<4>", "mat_transpose_f32x4_col2row")
_add("mat_transpose", "TR", "transpose 4x4 register blocks, 8x8 lanes per 32x32 block, 16 B and full lines both sides, no LDS (extents % 32; else the 1-D f32x4 rung)",
     "mat_transpose_f32x4_col2row2d", "mat_transpose_f32x4_row2col2d")
_add("mat_transpose", "TR", "transpose write-coalesced<1>", "mat_transpose_f32_row2col", "mat_transpose_f32_row2col2d")
_add("mat_transpose", "TR", "transpose write-coalesced<4>", "mat_transpose_f32x4_row2col")
_add("mat_transpose", "TR", "transpose write-coalesced<1>, diagonal block order", "mat_transpose_f32_diagonal2d")
_add("mat_transpose", "TR", "transpose 64x64 LDS tile, 16 B both sides", "mat_transpose_f32x4_shared_col2row2d",
     "mat_transpose_f32x4_shared_row2col2d")
_add("mat_transpose", "TR", "transpose 64x64 LDS tile padded (+1), 16 B both sides",
     "mat_transpose_f32x4_shared_bcf_col2row2d", "mat_transpose_f32x4_shared_bcf_row2col2d")

# ---------------------------------------------------------------- SGEMM (fp32)
#   S3 = G3 with fp32 tensors, S6 = G6 with fp32 tensors
_add("sgemm", "S3", "sgemm_naive(1 elem/thread)", "sgemm_naive_f32")
_add("sgemm", "S3", "sgemm_sliced_k(32x32x32 LDS)", "sgemm_sliced_k_f32")
_add("sgemm", "S3", "sgemm_valu_tile<BK=8,8x8,single buffer>", "sgemm_t_8x8_sliced_k_f32x4",
     "sgemm_t_8x8_sliced_k_f32x4_bcf", "sgemm_t_8x8_sliced_k_f32x4_bcf_offset")
_add("sgemm", "S3", "sgemm_valu_tile<BK=8,8x8,dbuf>", "sgemm_t_8x8_sliced_k_f32x4_bcf_dbuf",
     "sgemm_t_8x8_sliced_k_f32x4_bcf_dbuf_offset")
for _tn in (4, 8, 16):
    _add("sgemm", "S3", "sgemm_valu_tile<BK=16,8x%d,dbuf>" % _tn, "sgemm_t_8x%d_sliced_k16_f32x4_bcf_dbuf" % _tn)
    _add("sgemm", "S3", "sgemm_valu_tile<BK=16,8x%d,dbuf,issue-early/write-late>" % _tn,
         "sgemm_t_8x%d_sliced_k16_f32x4_bcf_dbuf_async" % _tn)
_add("sgemm", "S6", "sgemm_dma<64x128 | 128x128 | 256x128 by shape, 16-deep stages through a 3-slot LDS-DMA ring, v_mfma_f32_32x32x2_f32 (exact f32; no TF32 on gfx950)>",
     "sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages", "sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages_dsmem")
_add("sgemm_vendor", "S3", "rocblas_sgemm (exact f32)", "sgemm_cublas", "sgemm_cublas_tf32")

ENTRIES = tuple(_E)
BY_NAME = {e.name: e for e in ENTRIES}
assert len(BY_NAME) == len(ENTRIES), "duplicate exported name"

# which shared object holds which lib group
# comparison-row groups whose sources are optional at build time (_build.OPTIONAL_SOURCES): the vendor library may have been linked without them
OPTIONAL_LIBS = ("hgemm_vendor_lt", "fa2_vendor_ck")
SO_OF_LIB = {
    "hgemm": "libcln_amd.so", "flash_attn": "libcln_amd.so", "elementwise": "libcln_amd.so",
    "reduce": "libcln_amd.so", "softmax": "libcln_amd.so", "layer_norm": "libcln_amd.so",
    "rms_norm": "libcln_amd.so", "rope": "libcln_amd.so", "hgemm_vendor": "libcln_amd_vendor.so", "hgemm_vendor_lt": "libcln_amd_vendor.so", "fa2_vendor_ck": "libcln_amd_vendor.so",
    "histogram": "libcln_amd.so", "embedding": "libcln_amd.so", "activation": "libcln_amd.so",
    "sgemm": "libcln_amd.so", "sgemm_vendor": "libcln_amd_vendor.so",
    "dot_product": "libcln_amd.so", "sgemv": "libcln_amd.so", "hgemv": "libcln_amd.so", "mat_transpose": "libcln_amd.so",
}

# element dtype (torch name) each reduce rung takes, and the result dtype
REDUCE_DTYPES = {}
for e in ENTRIES:
    if e.sig == "R1":
        n = e.name[len("block_all_reduce_sum_"):]
        if n.startswith("f32"):
            REDUCE_DTYPES[e.name] = ("float32", "float32")
        elif n.startswith("f16"):
            REDUCE_DTYPES[e.name] = ("float16", "float32")
        elif n.startswith("bf16"):
            REDUCE_DTYPES[e.name] = ("bfloat16", "float32")
        elif n.startswith("fp8_e4m3"):
            REDUCE_DTYPES[e.name] = ("float8_e4m3fn", "float32")
        elif n.startswith("fp8_e5m2"):
            REDUCE_DTYPES[e.name] = ("float8_e5m2", "float32")
        else:
            REDUCE_DTYPES[e.name] = ("int8", "int32")


def entries_of(lib):
    return [e for e in ENTRIES if e.lib == lib]


# ---------------------------------------------------------------- name -> kernel for the run-time dispatched names
# (name, dims, stages) -> the text cln_describe() returns, i.e. what the dispatch code in csrc/ actually launches.
# dims = (M, N, K) for HGEMM names, (B, H, N, D) for flash-attn names. tests/test_describe.py asserts this table
# against the built library on a CPU-only box, so a change of the dispatch policy that is not reflected here fails CI.
_W4X2 = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
_SK = " (K %d per workgroup, fp32 partials in register layout) + hgemm_splitk_reduce [stages ignored: one pipeline]"  # more than 2 splits: the reduce launch
_SKF = " (K %d per workgroup, fp32 partials in register layout) + in-kernel fix-up by the last-arriving workgroup (one launch) [stages ignored: one pipeline]"  # 2 splits (round 5)
_SQKV = "flash_attn_mma_stages_split_q_shared_qkv"
_TQKV = "flash_attn_mma_stages_split_q_tiling_qkv"
_IGN = " [stages ignored: one pipeline]"
_ONE = " [single stage: every tile fetch waited for where it is issued]"
DISPATCH_EXAMPLES = [
    # HGEMM: BASELINE configs C2 (1024^3) and C3 (4096^3 / 8192^3) and the mid sizes
    (_W4X2, (4096, 4096, 4096), 2, "hgemm_w4<256x256x64,4 waves,128x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (4096, 4096, 4096), 0, "hgemm_pp<256x256x64,8 waves,4 slots,split DMA,LDS epilogue,NN>"),
    (_W4X2, (4096, 4096, 320), 2, "hgemm_pp<256x256x64,8 waves,4 slots,split DMA,LDS epilogue,NN>"),
    (_W4X2, (8192, 8192, 8192), 2, "hgemm_w4<256x256x64,4 waves,128x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (4096, 4096, 4096), 4, "hgemm_w4s<256x256,ring of 4 x 32-deep K slots,4 waves,128x128 wave tiles,LDS-DMA 3 slots ahead,LDS epilogue,NN>"),
    (_W4X2, (4096, 4096, 4096), 3, "hgemm_w4s<256x256,ring of 3 x 32-deep K slots,4 waves,128x128 wave tiles,LDS-DMA 2 slots ahead,LDS epilogue,NN>"),
    (_W4X2, (8192, 8192, 8192), 5, "hgemm_w4s<256x256,ring of 5 x 32-deep K slots,4 waves,128x128 wave tiles,LDS-DMA 4 slots ahead,LDS epilogue,NN>"),
    (_W4X2 + "_tn_swizzle_x4", (4096, 4096, 4096), 3, "hgemm_w4s<256x256,ring of 3 x 32-deep K slots,4 waves,128x128 wave tiles,LDS-DMA 2 slots ahead,LDS epilogue,TN>"),
    (_W4X2, (4096, 4096, 320), 4, "hgemm_pp32<256x256,BK=32 sub-tiles,4-deep ring,NN>"),  # K too short for the one-wave-per-SIMD kernels
    (_W4X2, (3072, 3072, 3072), 2, "hgemm_w4<192x192x64,4 waves,96x96 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (2304, 2304, 2304), 2, "hgemm_w4<192x192x64,4 waves,96x96 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (3072, 3072, 3136), 2, "hgemm_w4<192x192x64,4 waves,96x96 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (3072, 3072, 320), 2, "hgemm_pp<192x256x64,8 waves,4 slots,LDS epilogue,NN>"),
    (_W4X2, (6144, 6144, 6144), 2, "hgemm_w4<192x256x64,4 waves,96x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (4096, 6144, 4096), 2, "hgemm_w4<256x192x64,4 waves,128x96 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (3584, 3584, 3584), 2, "hgemm_w4<256x256x64,4 waves,128x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (2560, 2560, 2560), 2, "hgemm_w4<160x160x64,4 waves,80x80 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (2816, 2816, 2816), 2, "hgemm_w4<128x256x64,4 waves,64x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (3200, 3200, 3200), 2, "hgemm_w4<160x160x64,4 waves,80x80 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (4800, 4800, 4800), 2, "hgemm_w4<192x192x64,4 waves,96x96 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (4800, 4800, 320), 2, "mfma_ring<64x64x64,4 waves,stages=2,NN>"),
    (_W4X2, (2560, 2560, 2624), 2, "hgemm_w4<160x160x64,4 waves,80x80 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    (_W4X2, (2560, 2560, 320), 2, "mfma_ring<128x256x64,8 waves,stages=2,NN>"),
    (_W4X2, (1536, 1536, 1536), 2, "mfma_ring<64x64x64,4 waves,stages=2,NN>"),
    (_W4X2, (1792, 1792, 1792), 2, "mfma_ring<64x128x64,4 waves,stages=2,NN>"),
    (_W4X2, (2048, 2048, 2048), 2, "mfma_ring<64x128x64,4 waves,stages=2,NN>"),
    (_W4X2, (1024, 1024, 1024), 2, "mfma_ring<64x64x64,4 waves,stages=2,NN>"),
    # few output tiles, long K (K >= 4096, M N <= 2048^2): split-K over the same kernel, tile and number of splits from a fitted time model
    (_W4X2, (1024, 1024, 16384), 2, "hgemm_w4<128x256x64,4 waves,64x128 wave tiles,cross-tile LDS-DMA,NN> split-K x 8" + _SK % 2048),
    (_W4X2, (128, 8192, 8192), 3, "hgemm_w4<128x256x64,4 waves,64x128 wave tiles,cross-tile LDS-DMA,NN> split-K x 8" + _SK % 1024),
    (_W4X2, (2048, 2048, 16384), 2, "hgemm_w4<256x256x64,4 waves,128x128 wave tiles,cross-tile LDS-DMA,NN> split-K x 4" + _SK % 4096),
    (_W4X2, (768, 768, 12288), 2, "hgemm_w4<192x192x64,4 waves,96x96 wave tiles,cross-tile LDS-DMA,NN> split-K x 12" + _SK % 1024),
    (_W4X2 + "_tn_swizzle_x4", (640, 5120, 5120), 2, "hgemm_w4<160x160x64,4 waves,80x80 wave tiles,cross-tile LDS-DMA,TN> split-K x 2" + _SKF % 2560),
    # a few 256 x 256 tiles past whole rounds of 256: rows that fill whole rounds single-pass, the last tile rows split over K
    (_W4X2, (4352, 4352, 4352), 2, "hgemm_w4<256x256x64,4 waves,128x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN> on rows [0, 3840) + the last 2 tile rows as split-K x 4 (K 1088 per workgroup) + hgemm_splitk_reduce [tail split; stages ignored: one pipeline]"),
    (_W4X2 + "_tn_swizzle_x4", (7168, 7168, 7168), 4, "hgemm_w4<256x256x64,4 waves,128x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,TN> on rows [0, 6912) + the last 1 tile rows as split-K x 7 (K 1024 per workgroup) + hgemm_splitk_reduce [tail split; stages ignored: one pipeline]"),
    (_W4X2, (4608, 4608, 4608), 2, "hgemm_w4<192x256x64,4 waves,96x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),  # another tile shape already fills the rounds
    (_W4X2, (2048, 2048, 4096), 2, "mfma_ring<64x128x64,4 waves,stages=2,NN>"),      # M N > 1536^2 needs K >= 5120
    (_W4X2, (1024, 1024, 2048), 2, "mfma_ring<64x64x64,4 waves,stages=2,NN>"),       # K < 4096: single pass
    (_W4X2, (2560, 2560, 8192), 2, "hgemm_w4<160x160x64,4 waves,80x80 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),  # M N > 2048^2: single pass
    (_W4X2 + "_tn_swizzle_x4", (4096, 4096, 4096), 2, "hgemm_w4<256x256x64,4 waves,128x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,TN>"),
    ("hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem", (4096, 4096, 4096), 3, "mfma_ring<128x128x32,4 waves,stages=3,NN>"),
    ("hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem", (4096, 4096, 4096), 5, "mfma_ring<128x128x32,4 waves,stages=5,NN>"),
    ("hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem", (4096, 4096, 4096), 3, "hgemm_w4s<256x256,ring of 3 x 32-deep K slots,4 waves,128x128 wave tiles,LDS-DMA 2 slots ahead,LDS epilogue,NN>"),
    ("hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem", (4096, 4096, 128), 3, "mfma_ring<256x256x32,8 waves,stages=3,NN>"),  # fewer than 2 x 3 slots
    ("hgemm_mma_stages_block_swizzle_tn_cute", (4096, 4096, 4096), 2, "hgemm_w4<128x256x64,4 waves,64x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,TN>"),
    ("hgemm_mma_stages_block_swizzle_tn_cute", (4096, 4096, 4096), 3, "mfma_ring<128x256x64,8 waves,stages=3,TN>"),
    ("hgemm_mma_stages_block_swizzle_tn_cute", (4096, 4096, 4160), 2, "hgemm_w4<128x256x64,4 waves,64x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,TN>"),
    ("hgemm_mma_stages_block_swizzle_tn_cute", (4096, 4096, 320), 2, "mfma_ring<128x256x64,8 waves,stages=2,TN>"),
    ("hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem", (4096, 4096, 4096), 2, "hgemm_w4<256x256x64,4 waves,128x128 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    ("hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem", (4096, 4096, 4096), 2, "hgemm_w4<256x128x64,4 waves,128x64 wave tiles,cross-tile LDS-DMA,LDS epilogue,NN>"),
    ("hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem", (4096, 4096, 256), 2, "mfma_ring<256x128x64,8 waves,stages=2,NN>"),
    # flash-attn: config C4, its small-grid and D = 128 siblings, the stages knob, config C5 and the padded head dims
    ("flash_attn_mma_stages_split_kv", (4, 8, 2048, 64), 2,
     "fa2_fwd_splitkv<D=64,next K fragments prefetched into registers> 4 waves share 32 rows, 128-key tiles split over the waves, cross-wave max via LDS"),
    ("flash_attn_mma_stages_split_kv", (4, 8, 2048, 64), 1,
     "fa2_fwd_splitkv<D=64,load-then-compute> 4 waves share 32 rows, 128-key tiles split over the waves, cross-wave max via LDS"),
    (_SQKV, (4, 8, 2048, 64), 1, "fa2_fwd_m16x<D=64,BC=128,16x16x32 MFMA,pre-scaled Q,sum-checked softmax> 8 waves x 32 rows, two groups one phase apart" + _ONE),
    (_SQKV, (1, 48, 8192, 64), 1, "fa2_fwd_m16x64r<D=64,BC=64,16x16x32 MFMA,pre-scaled Q,sum-checked softmax> 8 waves x 64 rows, two groups one phase apart, K/V fragments shared by 4 query blocks" + _ONE),
    (_SQKV, (2, 32, 4096, 256), 1, "fa2_fwd_m16<D=256,BC=32,16x16x32 MFMA> 8 waves x 32 rows, two groups one phase apart" + _ONE),
    (_SQKV, (4, 8, 2048, 64), 2, "fa2_fwd_m16x<D=64,BC=128,16x16x32 MFMA,pre-scaled Q,sum-checked softmax> 8 waves x 32 rows, two groups one phase apart"),
    (_SQKV, (1, 48, 8192, 64), 2, "fa2_fwd_m16x64r<D=64,BC=64,16x16x32 MFMA,pre-scaled Q,sum-checked softmax> 8 waves x 64 rows, two groups one phase apart, K/V fragments shared by 4 query blocks"),
    (_SQKV, (2, 24, 4096, 64), 2, "fa2_fwd_m16x<D=64,BC=128,16x16x32 MFMA,pre-scaled Q,sum-checked softmax> 8 waves x 32 rows, two groups one phase apart"),
    (_SQKV, (4, 8, 2048, 128), 2, "fa2_fwd_m16x<D=128,BC=128,16x16x32 MFMA,pre-scaled Q,sum-checked softmax> 8 waves x 32 rows, two groups one phase apart"),
    (_SQKV, (2, 32, 4096, 256), 2, "fa2_fwd_m16<D=256,BC=32,16x16x32 MFMA> 8 waves x 32 rows, two groups one phase apart"),
    (_SQKV, (2, 8, 2048, 64), 2, "fa2_fwd_v2<D=64,NW=4,BC=64,prefetch,pre-scaled Q> 4 waves x 32 rows"),
    # the *_acc_f32 names at D <= 128: the same kernels with the scores scaled in fp32 (Q as loaded)
    (_SQKV + "_acc_f32", (4, 8, 2048, 64), 2, "fa2_fwd_m16x<D=64,BC=128,16x16x32 MFMA,fp32-scaled scores,sum-checked softmax> 8 waves x 32 rows, two groups one phase apart"),
    (_SQKV + "_acc_f32", (1, 48, 8192, 64), 1, "fa2_fwd_m16x64r<D=64,BC=64,16x16x32 MFMA,fp32-scaled scores,sum-checked softmax> 8 waves x 64 rows, two groups one phase apart, K/V fragments shared by 4 query blocks" + _ONE),
    (_SQKV + "_acc_f32", (2, 8, 2048, 128), 2, "fa2_fwd_m16x<D=128,BC=128,16x16x32 MFMA,fp32-scaled scores,sum-checked softmax> 8 waves x 32 rows, two groups one phase apart"),  # D = 128: the two-group kernel at every grid size
    (_SQKV + "_acc_f32", (2, 8, 1920, 128), 2, "fa2_fwd_v2<D=128,NW=4,BC=64,prefetch,fp32-scaled scores> 4 waves x 32 rows"),  # N a multiple of 128 only
    (_SQKV + "_acc_f32", (2, 32, 4096, 256), 2, "fa2_fwd_m16<D=256,BC=32,16x16x32 MFMA> 8 waves x 32 rows, two groups one phase apart"),
    (_SQKV, (1, 2, 256, 64), 2, "fa2_fwd_v2<D=64,NW=4,BC=64,prefetch,pre-scaled Q> 4 waves x 32 rows"),
    (_SQKV, (1, 40, 1024, 64), 2, "fa2_fwd_m16x<D=64,BC=128,16x16x32 MFMA,pre-scaled Q,sum-checked softmax> 8 waves x 32 rows, two groups one phase apart"),  # 160 workgroups of 256 rows
    (_SQKV, (1, 8, 1024, 256), 2, "fa2_fwd_m16<D=256,BC=32,16x16x32 MFMA> 8 waves x 32 rows, two groups one phase apart"),  # D = 256: at every grid size
    (_SQKV, (1, 8, 1152, 256), 2, "fa2_fwd_v2<D=256,NW=4,BC=64,prefetch> 4 waves x 32 rows"),
    (_SQKV, (1, 2, 192, 64), 1, "fa2_fwd_v2<D=64,NW=2,BC=64,load-then-compute,pre-scaled Q> 2 waves x 32 rows" + _ONE),
    (_SQKV + "_swizzle_qkv", (4, 8, 2048, 128), 2, "fa2_fwd_m16x<D=128,BC=128,16x16x32 MFMA,pre-scaled Q,sum-checked softmax,V^T> 8 waves x 32 rows, two groups one phase apart"),
    (_SQKV + "_swizzle_qkv", (2, 8, 2048, 128), 2, "fa2_fwd_m16x<D=128,BC=128,16x16x32 MFMA,pre-scaled Q,sum-checked softmax,V^T> 8 waves x 32 rows, two groups one phase apart"),
    (_SQKV + "_swizzle_qkv", (2, 8, 1920, 128), 2, "fa2_fwd_v2<D=128,NW=4,BC=64,prefetch,pre-scaled Q,V^T> 4 waves x 32 rows"),  # N a multiple of 128 only: the v2 kernel
    (_SQKV + "_swizzle_qkv", (4, 8, 2048, 128), 1, "fa2_fwd_m16x<D=128,BC=128,16x16x32 MFMA,pre-scaled Q,sum-checked softmax,V^T> 8 waves x 32 rows, two groups one phase apart" + _ONE),
    (_TQKV, (1, 32, 4096, 512), 2, "fa2_fwd_pair2<D=512,BC=32,16x16x32 MFMA,pairs of waves: rows split for QK^T and the softmax, d for PV> 8 waves, 128 rows, two groups one phase apart"),
    (_TQKV, (1, 16, 4096, 384), 2, "fa2_fwd_pair2<D=384,BC=32,16x16x32 MFMA,LDS geometry of D=512,pairs of waves: rows split for QK^T and the softmax, d for PV> 8 waves, 128 rows, two groups one phase apart"),
    (_TQKV, (1, 16, 4096, 320), 2, "fa2_fwd_pair2<D=320,BC=32,16x16x32 MFMA,LDS geometry of D=512,pairs of waves: rows split for QK^T and the softmax, d for PV> 8 waves, 128 rows, two groups one phase apart"),
    (_TQKV, (1, 16, 4096, 640), 2, "fa2_fwd_dw4<D=640,BC=16,2-slot K/V rings,O^T in AGPRs> 4 waves (one per SIMD) split d (160 columns each), 64 rows, softmax once per row by its owner wave"),
    (_TQKV, (1, 16, 4096, 1024), 2, "fa2_fwd_dw4<D=1024,BC=16,2-slot K/V rings,O^T in AGPRs> 4 waves (one per SIMD) split d (256 columns each), 64 rows, softmax once per row by its owner wave"),
    # stages = 1 above D = 256: the same kernel, single stage
    (_TQKV, (1, 32, 4096, 512), 1, "fa2_fwd_pair2<D=512,BC=32,16x16x32 MFMA,pairs of waves: rows split for QK^T and the softmax, d for PV> 8 waves, 128 rows, two groups one phase apart" + _ONE),
    (_TQKV, (1, 16, 4096, 1024), 1, "fa2_fwd_dw4<D=1024,BC=16,2-slot K/V rings,O^T in AGPRs> 4 waves (one per SIMD) split d (256 columns each), 64 rows, softmax once per row by its owner wave" + _ONE),
    (_TQKV, (1, 16, 4160, 768), 1, "fa2_fwd_dw4<D=768,BC=16,2-slot K/V rings,O^T in AGPRs> 4 waves (one per SIMD) split d (192 columns each), 64 rows, softmax once per row by its owner wave" + _ONE),
]


def describe(name, dims, stages=2):
    """Ask the built library which kernel `name` launches for `dims` (no GPU needed). Returns the text, or raises
    LookupError for a statically bound name / ValueError for an unsupported shape."""
    import ctypes
    from . import _loader
    lib = _loader.load_so("libcln_amd.so")
    buf = ctypes.create_string_buffer(512)
    d = list(dims) + [0] * (4 - len(dims))
    rc = lib.cln_describe(name.encode(), d[0], d[1], d[2], d[3], int(stages), buf, 512)
    if rc == -1:
        raise LookupError("%s is bound to one kernel: see manifest.BY_NAME[name].impl" % name)
    if rc < 0:
        raise ValueError("%s: shape %s not supported (status %d)" % (name, tuple(dims), rc))
    return buf.value.decode()


def stages_honoured(name, dims, stages=2):
    """True when `stages` selects the pipeline depth of the kernel `name` runs for `dims`, False when the plan has one pipeline and
    the value is ignored (cln_stages_honoured; the text of describe() then carries "stages ignored"). Same errors as describe()."""
    from . import _loader
    lib = _loader.load_so("libcln_amd.so")
    d = list(dims) + [0] * (4 - len(dims))
    rc = lib.cln_stages_honoured(name.encode(), d[0], d[1], d[2], d[3], int(stages))
    if rc == -1:
        raise LookupError("%s is bound to one kernel: see manifest.BY_NAME[name].impl" % name)
    if rc < 0:
        raise ValueError("%s: shape %s not supported (status %d)" % (name, tuple(dims), rc))
    return bool(rc)
