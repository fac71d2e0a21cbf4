"""FlashAttention-2 forward bench driver -- same CLI flags, row tags, FLOP model, tolerance and printed
columns as reference kernels/flash-attn/flash_attn_mma.py (flags :21-48, rows :526-592, --check :596-701,
get_qkvo :353-380), re-authored as a table generated from the tag grammar.

  python flash_attn_mma.py --B 4 --H 8 --N 2048 --D 64 --check       # BASELINE config C4
  python flash_attn_mma.py --B 1 --H 32 --N 4096 --D 512 --sdpa      # config C5

The `(flash)` row of the reference (flash_attn pip package) is replaced by torch SDPA on ROCm, which is
the check target here for every D (the reference uses SDPA only for D > 256).
"""
import argparse
import math
import os
import random
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, emit_json, package, sync  # noqa: E402

torch.set_printoptions(precision=6, threshold=8, edgeitems=3, linewidth=120, sci_mode=False)


def get_args():
    p = argparse.ArgumentParser()
    p.add_argument("--no-rand-q", "--no-rq", action="store_true")
    p.add_argument("--no-rand-k", "--no-rk", action="store_true")
    p.add_argument("--no-rand-v", "--no-rv", action="store_true")
    p.add_argument("--no-rand-qkv", "--no-rqkv", action="store_true")
    p.add_argument("--run-torch-unfused", "--torch", action="store_true")
    p.add_argument("--run-torch-sdpa", "--sdpa", action="store_true")
    p.add_argument("--check", action="store_true")
    p.add_argument("--check-all", action="store_true")
    p.add_argument("--show-all", "--show", action="store_true")
    p.add_argument("--show-matrix", action="store_true")
    p.add_argument("--only-flops-matmul", "--flops-mm", action="store_true")
    p.add_argument("--run-acc-f32", "--acc-f32", "--f32", action="store_true")
    for d in ("--B", "--H", "--N", "--D", "--seed"):
        p.add_argument(d, type=int, default=None)
    p.add_argument("--sleep", type=float, default=0.05)
    p.add_argument("--debug", action="store_true")
    p.add_argument("--verbose", "--v", action="store_true")
    p.add_argument("--warmup", "--w", type=int, default=1)
    p.add_argument("--iters", "--i", type=int, default=5)
    p.add_argument("--range-k", "--gk", action="store_true")
    p.add_argument("--build-others", "--others", action="store_true")
    p.add_argument("--tag-hints", "--tags", "--hints", type=str, default=None)
    return p.parse_args()


def rows_table():
    """(tag, function suffix, V transposed?) in the reference's order (flash_attn_mma.py:528-589)."""
    P = "flash_attn_mma_stages_"
    rows = []

    def fam(tag, fn, vt=False):
        for st in (1, 2):
            rows.append(("mma(%s+stage%d)" % (tag, st), P + fn, st, vt))

    fam("split-kv", "split_kv")
    fam("split-q", "split_q")
    fam("split-q+share-kv", "split_q_shared_kv")
    fam("split-q+share-kv+acc-f32", "split_q_shared_kv_acc_f32")
    for s in ("q", "qk", "qkv"):
        fam("split-q+share-kv+swizzle-" + s, "split_q_shared_kv_swizzle_" + s, s == "qkv")
    fam("split-q+share-qkv", "split_q_shared_qkv")
    fam("split-q+share-qkv+acc-f32", "split_q_shared_qkv_acc_f32")
    for s in ("q", "qk", "qkv"):
        fam("split-q+share-qkv+swizzle-" + s, "split_q_shared_qkv_swizzle_" + s, s == "qkv")
    fam("split-q+tiling-qk", "split_q_tiling_qk")
    fam("split-q+tiling-qk+acc-f32", "split_q_tiling_qk_acc_f32")
    for s in ("q", "qk", "qkv"):
        fam("split-q+tiling-qk+swizzle-" + s, "split_q_tiling_qk_swizzle_" + s, s == "qkv")
    fam("split-q+tiling-qkv", "split_q_tiling_qkv")
    for s in ("q", "qk", "qkv"):
        fam("split-q+tiling-qkv+swizzle-" + s, "split_q_tiling_qkv_swizzle_" + s)
    fam("split-q+tiling-qkv+acc-f32", "split_q_tiling_qkv_acc_f32")
    for s in ("q", "qk", "qkv"):
        fam("split-q+tiling-qkv+acc-f32+swizzle-" + s, "split_q_tiling_qkv_acc_f32_swizzle_" + s)
    fam("split-q+share-kv+acc-f32+rr", "split_q_shared_kv_acc_f32_rr")
    fam("split-q+share-qkv+o-s2g", "split_q_shared_qkv_Os2g")
    fam("split-q+share-qkv+acc-f32+rr", "split_q_shared_qkv_acc_f32_rr")
    return rows


def max_headdim(fname, stages, manifest):
    """Reference MAX_HEADDIM_CFG (:436-506): share-kv/share-qkv stage 2 rows stop at 128."""
    d = manifest.FA_MAX_HEADDIM[fname]
    if stages == 2 and d == 256 and "tiling" not in fname and not fname.endswith("acc_f32_rr"):
        return 128
    if stages == 2 and fname.endswith("shared_kv_acc_f32_rr"):
        return 128
    return d


def get_qkvo(args, B, H, N, D):
    def mk(no_rand):
        return (torch.ones if (no_rand or args.no_rand_qkv) else torch.randn)(B, H, N, D, device=DEVICE,
                                                                              dtype=torch.half).contiguous()
    q, k, v = mk(args.no_rand_q), mk(args.no_rand_k), mk(args.no_rand_v)
    if args.range_k:  # K row i = (i + 1) / N
        k = ((torch.arange(N, device=DEVICE, dtype=torch.float32) + 1) / N).half().view(1, 1, N, 1).expand(B, H, N, D).contiguous()
    o = torch.zeros(B, H, N, D, device=DEVICE, dtype=torch.half).contiguous()
    tv = v.transpose(-2, -1).contiguous()
    return q, k, v, o, tv


def unfused_standard_attn(q, k, v):
    att = (q @ k.transpose(-2, -1) * (1.0 / math.sqrt(k.size(-1))))
    return F.softmax(att, dim=-1) @ v


def main():
    args = get_args()
    from cuda_learn_notes_amd import bench_utils as bu, manifest
    if not HAS_GPU:
        sys.exit("flash_attn_mma.py: no GPU visible; the kernel rows have no CPU path "
                 "(the CPU oracle lives in oracle/ and is exercised by tests/)")
    lib = package().flash_attn_lib()
    seed = args.seed if args.seed else random.choice(range(10000))
    torch.manual_seed(seed)
    random.seed(seed)
    Bs = [1, 4, 8] if not args.B else [args.B]
    Hs = [1, 4, 8] if not args.H else [args.H]
    Ns = [1024, 2048, 4096, 8192] if not args.N else [args.N]
    Ds = [64, 128] if not args.D else [args.D]
    hints = [h for h in (args.tag_hints or "").strip().split(",") if h]
    json_rows = []
    for B in Bs:
        for H in Hs:
            for N in Ns:
                for D in Ds:
                    bu.pretty_print_line()
                    bu.pretty_print_line(f"B={B}, H={H}, N={N}, D={D}, Warmup: {args.warmup}, Iters: {args.iters}", " ")
                    bu.pretty_print_line()
                    q, k, v, o, tv = get_qkvo(args, B, H, N, D)
                    state = {"max": -1.0}
                    outs = {}

                    def bench(tag, call, always=False):
                        for _ in range(args.warmup):
                            out = call()
                        sync()
                        t0 = time.time()
                        for _ in range(args.iters):
                            out = call()
                        sync()
                        secs = (time.time() - t0) / args.iters
                        tfl = bu.get_mha_tflops(B, H, N, D, secs, only_matmul=args.only_flops_matmul)
                        flat = out.flatten()
                        vals = [f"{round(x, 8):<12}" for x in (flat[0].item(), flat[1].item(), flat[-1].item())]
                        line = f"{tag:>50}: {vals}, time:{str(secs * 1e3)[:8]}ms, TFLOPS:{tfl:<6.2f}"
                        if tfl > state["max"]:
                            imp = 0 if state["max"] <= 0 else round((tfl - state["max"]) / state["max"] * 100, 2)
                            state["max"] = tfl
                            print(line + f"(+{imp:.2f}%)")
                        elif args.show_all or always:
                            print(line)
                        if args.show_matrix:
                            print(out)
                        json_rows.append({"kernel": tag, "shape": [B, H, N, D], "ms": secs * 1e3, "tflops": tfl,
                                          "tflops_4bhn2d": bu.mha_flops_conventional(B, H, N, D) / secs * 1e-12,
                                          "roofline": {"bound": "mfma", "peak": bu.PEAK_FP16_MFMA_TFLOPS,
                                                       "achieved": tfl, "frac": tfl / bu.PEAK_FP16_MFMA_TFLOPS}})
                        time.sleep(args.sleep)
                        return out.clone()

                    if args.run_torch_unfused:
                        outs["(unfused)"] = bench("(unfused)", lambda: unfused_standard_attn(q, k, v))
                    for tag, fname, stages, vt in rows_table():
                        if hints and not any(h in tag for h in hints):
                            continue
                        if not args.build_others and ("s2g" in tag or "rr" in tag):
                            continue
                        if "acc-f32" in tag and not args.run_acc_f32:
                            continue
                        if D > max_headdim(fname, stages, manifest):
                            continue
                        fn = getattr(lib, fname)
                        o.fill_(0)
                        vv = tv if vt else v
                        try:
                            outs[tag] = bench(tag, lambda: (fn(q, k, vv, o, stages), o)[1])
                        except RuntimeError as e:
                            print(f"{tag:>50}: skipped ({e})")
                    out_sdpa = bench("(sdpa)", lambda: F.scaled_dot_product_attention(q, k, v), always=True) \
                        if (args.run_torch_sdpa or args.check or D > 256) else None
                    bu.pretty_print_line()
                    if args.check and out_sdpa is not None:
                        for tag, out in outs.items():
                            if tag == "(unfused)":
                                continue
                            diff = (out_sdpa - out).abs()
                            ok = str(torch.allclose(out_sdpa, out, atol=1e-2))
                            bu.pretty_print_line(
                                f"out_sdpa vs {tag:<42}, all close: {ok:<6}, max diff: {diff.max().item():.6f}, "
                                f"min diff: {diff.min().item():.6f}, mean diff: {diff.mean().item():.6f}")
                        bu.pretty_print_line()
    emit_json(json_rows)


if __name__ == "__main__":
    main()
