"""GPU sweep: times every HGEMM ring variant, the reference-named rungs, rocBLAS, the FA2 kernels and
the bandwidth kernels; dumps one JSON to gpurun_out/. Run on the GPU box:
    python cuda-learn-notes_amd/tools/tune_gpu.py [--quick]
"""
import argparse
import json
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tune.json"))
    args = ap.parse_args()
    pkg = entry.load_package()
    from cuda_learn_notes_amd import bench_utils as bu, host
    dev = torch.device("cuda:0")
    res = {"device": torch.cuda.get_device_name(0), "hgemm": [], "fa": [], "bw": []}
    hg = pkg.hgemm_lib()
    hg.init_cublas_handle()

    def ev(fn, w=3, it=10):
        ms, mn, _ = bu.time_call_events(fn, w, it)
        return ms, mn

    sizes = [4096] if args.quick else [1024, 2048, 4096, 8192]
    for S in sizes:
        M = N = K = S
        torch.manual_seed(S)
        a = torch.randn(M, K, dtype=torch.half, device=dev)
        b = torch.randn(K, N, dtype=torch.half, device=dev)
        bt = bu.as_col_major(b)
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        fl = bu.hgemm_flops(M, N, K)
        stride = bu.make_block_swizzle_stride(N, K)

        def rec(tag, fn):
            try:
                ms, mn = ev(fn)
                res["hgemm"].append({"size": S, "tag": tag, "ms": ms, "min_ms": mn, "tflops": fl / ms * 1e-9,
                                     "tflops_best": fl / mn * 1e-9})
                print("%5d %-48s %8.4f ms %8.1f TF (best %8.1f)" % (S, tag, ms, fl / ms * 1e-9, fl / mn * 1e-9),
                      flush=True)
            except Exception as e:
                res["hgemm"].append({"size": S, "tag": tag, "error": str(e)[:200]})
                print(S, tag, "ERR", str(e)[:120], flush=True)

        rec("rocblas_nn", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c))
        rec("rocblas_tn", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c))
        rec("torch.matmul", lambda: torch.matmul(a, b, out=c))
        for layout, bb in ((0, b), (1, bt)):
            for tile in (0, 1, 2, 3):
                for bk in (64, 32):
                    for st in (2, 3, 4, 5):
                        for swz, sstride in ((0, 1), (1, stride), (1, 1024)):
                            if swz and sstride == 1024 and (S < 4096 or tile != 1):
                                continue
                            tag = "ring L%d T%d bk%d s%d swz%d/%d" % (layout, tile, bk, st, swz, sstride)
                            rec(tag, lambda: host.hgemm_variant(0, layout, tile, bk, st, a, bb, c, swz, sstride))
        rec("1stage 128x128x32 NN", lambda: host.hgemm_variant(1, 0, 0, 32, 1, a, b, c))
        if S <= 4096:
            rec("naive mfma NN", lambda: host.hgemm_variant(2, 0, 0, 16, 1, a, b, c))
            rec("valu t8x8 k16 dbuf async", lambda: hg.hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf_async(a, b, c))
            rec("valu t8x8 k32 dbuf", lambda: hg.hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf(a, b, c))
            rec("valu t16x8 k32 dbuf async", lambda: hg.hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf_async(a, b, c))
            rec("valu t8x8 k8", lambda: hg.hgemm_t_8x8_sliced_k_f16x8_pack_bcf(a, b, c))
        if S <= 2048:
            rec("valu naive", lambda: hg.hgemm_naive_f16(a, b, c))
            rec("valu sliced_k", lambda: hg.hgemm_sliced_k_f16(a, b, c))
        del a, b, bt, c
    hg.destroy_cublas_handle()

    # ---- flash attention
    fa = pkg.flash_attn_lib()
    fa_shapes = [(4, 8, 2048, 64), (4, 8, 2048, 128), (1, 8, 8192, 64), (1, 48, 8192, 64), (4, 8, 2048, 32),
                 (2, 8, 2048, 256), (1, 32, 4096, 512), (1, 8, 8192, 512), (1, 8, 2048, 1024)]
    if args.quick:
        fa_shapes = fa_shapes[:2] + [(1, 32, 4096, 512)]
    for (B, H, N, D) in fa_shapes:
        q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
        o = torch.zeros_like(q)
        vt = v.transpose(-2, -1).contiguous()
        cands = [("shared_qkv", fa.flash_attn_mma_stages_split_q_shared_qkv, v)] if D <= 256 else \
                [("tiling_qkv", fa.flash_attn_mma_stages_split_q_tiling_qkv, v)]
        if D <= 256:
            cands.append(("shared_qkv_swizzle_qkv(Vt)", fa.flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv, vt))
        for tag, fn, vv in cands:
            for st in (1, 2):
                try:
                    ms, mn = ev(lambda: fn(q, k, vv, o, st), 3, 10)
                    tf = bu.mha_flops_conventional(B, H, N, D) / ms * 1e-9
                    res["fa"].append({"shape": [B, H, N, D], "tag": tag, "stages": st, "ms": ms, "min_ms": mn,
                                      "tflops_4bhn2d": tf, "tflops_ref_model": bu.get_mha_tflops(B, H, N, D, ms * 1e-3)})
                    print("FA %s %-28s s%d %8.4f ms %8.1f TF" % ((B, H, N, D), tag, st, ms, tf), flush=True)
                except Exception as e:
                    res["fa"].append({"shape": [B, H, N, D], "tag": tag, "stages": st, "error": str(e)[:200]})
                    print("FA", (B, H, N, D), tag, "ERR", str(e)[:120], flush=True)
        try:
            import torch.nn.functional as F
            ms, mn = ev(lambda: F.scaled_dot_product_attention(q, k, v), 3, 10)
            tf = bu.mha_flops_conventional(B, H, N, D) / ms * 1e-9
            res["fa"].append({"shape": [B, H, N, D], "tag": "torch sdpa", "ms": ms, "tflops_4bhn2d": tf})
            print("FA %s %-28s    %8.4f ms %8.1f TF" % ((B, H, N, D), "torch sdpa", ms, tf), flush=True)
        except Exception as e:
            print("sdpa ERR", str(e)[:100])
        del q, k, v, o, vt

    # ---- bandwidth kernels at the script shapes (S=K=4096)
    lib = pkg.load("elementwise", "reduce", "softmax", "layer_norm", "rms_norm", "rope")
    S = K = 4096
    x = torch.randn(S, K, device=dev)
    y = torch.randn(S, K, device=dev)
    z = torch.zeros(S, K, device=dev)
    xh, yh, zh = x.half(), y.half(), z.half()

    def bw(tag, fn, nbytes):
        try:
            ms, mn = ev(fn, 5, 30)
            res["bw"].append({"tag": tag, "ms": ms, "min_ms": mn, "gbps": nbytes / ms * 1e-6})
            print("BW %-44s %8.4f ms %8.1f GB/s" % (tag, ms, nbytes / ms * 1e-6), flush=True)
        except Exception as e:
            res["bw"].append({"tag": tag, "error": str(e)[:200]})
            print("BW", tag, "ERR", str(e)[:120], flush=True)

    n = S * K
    for nm in ("elementwise_add_f32", "elementwise_add_f32x4"):
        bw(nm, lambda nm=nm: getattr(lib, nm)(x, y, z), 12 * n)
    bw("torch.add f32", lambda: torch.add(x, y, out=z), 12 * n)
    for nm in ("elementwise_add_f16", "elementwise_add_f16x2", "elementwise_add_f16x8", "elementwise_add_f16x8_pack"):
        bw(nm, lambda nm=nm: getattr(lib, nm)(xh, yh, zh), 6 * n)
    for nm, t, eb in (("f32_f32", x, 4), ("f32x4_f32", x, 4), ("f16_f32", xh, 2), ("f16x8_pack_f32", xh, 2),
                      ("f16x8_pack_f16", xh, 2)):
        bw("reduce " + nm, lambda nm=nm, t=t: getattr(lib, "block_all_reduce_sum_" + nm)(t), eb * n)
    bw("torch.sum f32", lambda: torch.sum(x), 4 * n)
    xi = torch.randint(-128, 127, (S, K), dtype=torch.int8, device=dev)
    bw("reduce i8x16_pack_i32", lambda: lib.block_all_reduce_sum_i8x16_pack_i32(xi), n)
    for nm in ("softmax_f32_per_token", "safe_softmax_f32x4_per_token", "online_safe_softmax_f32x4_pack_per_token"):
        bw(nm, lambda nm=nm: getattr(lib, nm)(x, z), 8 * n)
    bw("safe_softmax_f16x8_pack_f32_per_token", lambda: lib.safe_softmax_f16x8_pack_f32_per_token(xh, zh), 4 * n)
    bw("torch.softmax f32", lambda: torch.softmax(x, dim=1, out=z), 8 * n)
    for nm in ("layer_norm_f32", "layer_norm_f32x4"):
        bw(nm, lambda nm=nm: getattr(lib, nm)(x, z, 1.0, 0.0), 8 * n)
    bw("layer_norm_f16x8_pack_f32", lambda: lib.layer_norm_f16x8_pack_f32(xh, zh, 1.0, 0.0), 4 * n)
    bw("rms_norm_f32x4", lambda: lib.rms_norm_f32x4(x, z, 1.0), 8 * n)
    bw("rms_norm_f16x8_pack_f16", lambda: lib.rms_norm_f16x8_pack_f16(xh, zh, 1.0), 4 * n)
    xr = torch.randn(8192, 1024, device=dev)
    zr = torch.zeros_like(xr)
    bw("rope_f32", lambda: lib.rope_f32(xr, zr), 8 * xr.numel())
    bw("rope_f32x4_pack", lambda: lib.rope_f32x4_pack(xr, zr), 8 * xr.numel())

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    try:
        main()
    except Exception:
        traceback.print_exc()
        sys.exit(1)
