"""FlashAttention-2 forward bench driver -- same CLI flags, row tags, FLOP model, tolerance, printed columns AND
helper functions as reference kernels/flash-attn/flash_attn_mma.py: get_args :21-48, get_mha_tflops :191-222,
run_benchmark :229-347 (same parameters, returns (out.clone(), mean_time_ms) or (None, None) for a skipped row),
get_qkvo :350-381, unfused_standard_attn :384-388, sdpa :391-398, check_all_close :401-427, MAX_HEADDIM_CFG :436-506,
rows :526-592 (generated here from the tag grammar). `from flash_attn_mma import check_all_close, run_benchmark` works
without a GPU; the kernel rows need one (no CPU fallback).

  python flash_attn_mma.py --B 4 --H 8 --N 2048 --D 64 --check       # BASELINE config C4
  python flash_attn_mma.py --B 1 --H 32 --N 4096 --D 512 --sdpa      # config C5

The `(flash)` row of the reference (flash_attn pip package, not in the ROCm image) is replaced by two rows: `(ck_tile fmha)`
-- AMD's ck_tile FMHA forward kernels, what that package's ROCm backend dispatches to, compiled from the image's headers
(D = 64 / 128) -- and torch SDPA, which is the check target here for every D (the reference uses SDPA only for D > 256).
"""
import argparse
from functools import partial
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


def get_args(argv=None):
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
    return p.parse_args(argv)


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


from cuda_learn_notes_amd import bench_utils as bu, manifest  # noqa: E402

get_mha_tflops = bu.get_mha_tflops
pretty_print_line = bu.pretty_print_line
# the reference parses its flags at import (:50); imported as a module (tests) this uses the defaults
args = get_args(None if __name__ == "__main__" else [])

MAX_TFLOPS = -1
JSON_ROWS = []


def max_headdim(fname, stages):
    """Reference MAX_HEADDIM_CFG (:436-506): share-kv/share-qkv stage 2 rows stop at 128."""
    d = manifest.FA_MAX_HEADDIM[fname]
    if stages == 2 and d == 256 and "tiling" not in fname and not fname.endswith("acc_f32_rr"):
        return 128
    if stages == 2 and fname.endswith("shared_kv_acc_f32_rr"):
        return 128
    return d


# tag -> max head dim, the table the reference keeps by hand (:436-506)
MAX_HEADDIM_CFG = {"(flash)": 256, "(sdpa)": 4096, "(unfused)": 4096}
MAX_HEADDIM_CFG.update({tag: max_headdim(fname, st) for tag, fname, st, _ in rows_table()})


def run_benchmark(perf_func, q, k, v, tag, out=None, s=None, stages=-1, warmup=None, iters=None, show_matrix=None,
                  only_show_improved=None):
    """Time one attention row with the reference protocol and print it in the reference format; rows filtered out by
    --tag-hints / --build-others / --sdpa / --torch / --acc-f32 or by the head-dim table return (None, None).
    `perf_func(q, k, v, out, stages)` for the kernel rows, `perf_func(q, k, v)` for the torch rows."""
    global MAX_TFLOPS
    warmup = args.warmup if warmup is None else warmup
    iters = args.iters if iters is None else iters
    show_matrix = args.show_matrix if show_matrix is None else show_matrix
    only_show_improved = (not args.show_all) if only_show_improved is None else only_show_improved
    if args.tag_hints:
        hints = args.tag_hints.strip().split(",") + ["flash", "sdpa", "unfused", "ck_tile"]
        if not any(h in tag for h in hints):
            return None, None
    if not args.build_others and any(t in tag for t in ("s2g", "rr")):
        return None, None
    if "sdpa" in tag and not args.run_torch_sdpa:
        return None, None
    if "unfused" in tag and not args.run_torch_unfused:
        return None, None
    if "acc-f32" in tag and not args.run_acc_f32:
        return None, None
    B, H, N, D = q.size()
    if "flash" in tag:
        B, N, H, D = q.size()
    if D > MAX_HEADDIM_CFG.get(tag, 1 << 30):
        return None, None
    if out is not None:
        out.fill_(0)
    if s is not None:
        s.fill_(0)

    def call():
        if out is None:
            return perf_func(q, k, v)
        if stages >= 1:
            if s is not None:
                perf_func(q, k, v, out, s, stages)
            else:
                perf_func(q, k, v, out, stages)
        else:
            perf_func(q, k, v, out)
        return out

    res = None
    for _ in range(warmup):
        res = call()
    sync()
    start = time.time()
    for _ in range(iters):
        res = call()
    sync()
    mean_secs = (time.time() - start) / iters
    mean_time = mean_secs * 1000
    out = res
    TFLOPS = get_mha_tflops(B, H, N, D, mean_secs, only_matmul=args.only_flops_matmul)
    flat = out.flatten()
    out_val = [f"{round(x, 8):<12}" for x in (flat[0].item(), flat[1].item(), flat[-1].item())]
    line = f"{tag:>50}: {out_val}, time:{str(mean_time)[:8]}ms, TFLOPS:{TFLOPS:<6.2f}"
    if TFLOPS > MAX_TFLOPS:
        improve = round((TFLOPS - MAX_TFLOPS) / MAX_TFLOPS * 100, 2) if MAX_TFLOPS > 0 else 0
        MAX_TFLOPS = TFLOPS
        print(line + f"(+{improve:.2f}%)")
    elif (not only_show_improved) or ("flash" in tag) or ("sdpa" in tag) or ("ck_tile" in tag):
        print(line)
    if show_matrix:
        print(out)
    JSON_ROWS.append({"kernel": tag, "shape": [B, H, N, D], "ms": mean_time, "tflops": TFLOPS,
                      "tflops_4bhn2d": bu.mha_flops_conventional(B, H, N, D) / mean_secs * 1e-12,
                      "roofline": {"bound": "mfma", "peak": bu.PEAK_FP16_MFMA_TFLOPS, "achieved": TFLOPS,
                                   "frac": TFLOPS / bu.PEAK_FP16_MFMA_TFLOPS}})
    time.sleep(args.sleep)
    sync()
    return out.clone(), mean_time


def get_qkvo(B, H, N, D):
    """q, k, v, o as [B,H,N,D] fp16 plus the layouts the reference rows take: fq/fk/fv = [B,N,H,D] (flash-attn
    package), tk/tv = [B,H,D,N] (the *_swizzle_qkv rows take V transposed). Reference :350-381."""
    def mk(no_rand):
        return (torch.ones if (no_rand or args.no_rand_qkv) else torch.randn)(B, H, N, D, device=DEVICE,
                                                                              dtype=torch.half).contiguous()
    q, k, v = mk(args.no_rand_q), mk(args.no_rand_k), mk(args.no_rand_v)
    if args.range_k and (args.no_rand_k or args.no_rand_qkv):  # K row i = (i + 1) / N
        k = ((torch.arange(N, device=DEVICE, dtype=torch.float32) + 1) / N).half().view(1, 1, N, 1).expand(B, H, N, D).contiguous()
    o = torch.zeros(B, H, N, D, device=DEVICE, dtype=torch.half).contiguous()
    fq, fk, fv = (t.transpose(1, 2).contiguous() for t in (q, k, v))
    tk, tv = k.transpose(-2, -1).contiguous(), v.transpose(-2, -1).contiguous()
    return q, k, v, o, fq, fk, fv, tk, tv


def unfused_standard_attn(q, k, v):
    att = (q @ k.transpose(-2, -1) * (1.0 / math.sqrt(k.size(-1))))
    return F.softmax(att, dim=-1) @ v


def sdpa(q, k, v, use_flash: bool = False):
    """torch SDPA with the backend forced as the reference does (:391-398): memory-efficient by default, flash on
    request. On PyTorch-ROCm both map to the AOTriton / CK kernels; falls back to the default dispatch when the forced
    backend is unavailable for the shape."""
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel
        with sdpa_kernel(SDPBackend.FLASH_ATTENTION if use_flash else SDPBackend.EFFICIENT_ATTENTION):
            return F.scaled_dot_product_attention(q, k, v)
    except (ImportError, RuntimeError):
        return F.scaled_dot_product_attention(q, k, v)


def check_all_close(out_flash_or_sdpa, out_mma, tag: str = "out_mma", check_all: bool = False, is_flash: bool = True):
    """Print `all close` (atol 1e-2, the reference tolerance :421) and the max / min / mean |diff| of one row against
    the flash (given as [B,N,H,D]) or SDPA (given as [B,H,N,D]) output; silently skips rows that did not run.
    Returns the allclose verdict (the reference returns None; tests use the value)."""
    if out_flash_or_sdpa is None or out_mma is None:
        return None
    true_tag = "out_flash" if is_flash else "out_sdpa"
    if is_flash:
        out_flash_or_sdpa = out_flash_or_sdpa.transpose(1, 2)
    if check_all:
        for i in range(min(4, out_mma.size(2) // 8)):
            pretty_print_line()
            print(f"{true_tag}[:, :,  {i * 8}:{(i + 1) * 8}, :]:\n")
            print(out_flash_or_sdpa[:, :, i * 8:(i + 1) * 8, :].float())
            print(f"{tag}[:, :, {i * 8}:{(i + 1) * 8}, :]:\n")
            print(out_mma[:, :, i * 8:(i + 1) * 8, :].float())
        pretty_print_line()
    diff = torch.abs(out_flash_or_sdpa - out_mma)
    ok = bool(torch.allclose(out_flash_or_sdpa, out_mma, atol=1e-2))
    pretty_print_line(f"{true_tag} vs {tag:<25}, all close: {str(ok):<6}, max diff: {diff.max().item():.6f}, "
                      f"min diff: {diff.min().item():.6f}, mean diff: {diff.mean().item():.6f}")
    return ok


CK_FMHA = None  # set by main(): AMD's ck_tile FMHA forward through libcln_amd_vendor.so


def main():
    global MAX_TFLOPS, CK_FMHA
    if not HAS_GPU:
        sys.exit("flash_attn_mma.py: no GPU visible; the kernel rows have no CPU path "
                 "(the CPU oracle lives in oracle/ and is exercised by tests/)")
    lib = package().flash_attn_lib()
    try:  # the vendor comparison row; a missing vendor library never stops the kernel rows
        CK_FMHA = package().load("fa2_vendor_ck").cln_fa2_ck_tile_fwd
    except Exception:
        CK_FMHA = None
    seed = args.seed if args.seed else random.choice(range(10000))
    torch.manual_seed(seed)
    random.seed(seed)
    Bs = [1, 4, 8] if not args.B else [args.B]
    Hs = [1, 4, 8] if not args.H else [args.H]
    Ns = [1024, 2048, 4096, 8192] if not args.N else [args.N]
    Ds = [64, 128, 256, 512] if not args.D else [args.D]
    for B in Bs:
        for H in Hs:
            for N in Ns:
                for D in Ds:
                    MAX_TFLOPS = -1
                    q, k, v, o, fq, fk, fv, tk, tv = get_qkvo(B, H, N, D)
                    sync()
                    pretty_print_line()
                    pretty_print_line(f"B={B}, H={H}, N={N}, D={D}, Warmup: {args.warmup}, Iters: {args.iters}", " ")
                    pretty_print_line()
                    outs = {}
                    run_benchmark(unfused_standard_attn, q, k, v, "(unfused)")
                    for tag, fname, stages, vt in rows_table():
                        try:
                            outs[tag], _ = run_benchmark(getattr(lib, fname), q, k, tv if vt else v, tag, o, stages=stages)
                        except RuntimeError as e:
                            print(f"{tag:>50}: skipped ({e})")
                    # the `(flash)` row of the reference is flash_attn_func (:591). The flash_attn pip package is not in the
                    # ROCm image, but the kernels its ROCm backend dispatches to are: AMD's ck_tile FMHA forward, compiled from
                    # the image's headers into the vendor library (csrc/fa2_vendor_ck.hip; D = 64 / 128)
                    if D in (64, 128) and CK_FMHA is not None:
                        try:
                            outs["(ck_tile fmha)"], _ = run_benchmark(lambda a, b, c, out: CK_FMHA(a, b, c, out, 0), q, k, v,
                                                                      "(ck_tile fmha)", o)
                        except RuntimeError as e:
                            print(f"{'(ck_tile fmha)':>50}: skipped ({e})")
                    # ... and torch SDPA for every D (the reference itself switches to SDPA for D > 256)
                    run_sdpa = args.run_torch_sdpa
                    args.run_torch_sdpa = run_sdpa or args.check or D > 256
                    out_sdpa, _ = run_benchmark(partial(sdpa, use_flash=(D <= 256)), q, k, v, "(sdpa)")
                    args.run_torch_sdpa = run_sdpa
                    pretty_print_line()
                    if args.check:
                        for tag, out in outs.items():
                            check_all_close(out_sdpa, out, "out_" + tag.replace("mma(", "mma_").rstrip(")"),
                                            args.check_all, is_flash=False)
                        pretty_print_line()
    emit_json(JSON_ROWS)


if __name__ == "__main__":
    main()
