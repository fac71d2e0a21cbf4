"""HGEMM bench driver -- same CLI flags, row tags and printed columns as reference
kernels/hgemm/hgemm.py (flags :16-52, tag table :320-421, print format :141-168), re-authored as a table.

  python hgemm.py --mma --MNK 4096          # BASELINE config C3
  python hgemm.py --mma-all --wmma-all --cuda-all --mma-tn --cute-tn --torch

Rows whose tag contains "cublas" run the rocBLAS row (libcln_amd_vendor.so). Each row also goes to the
JSON side channel ($CLN_AMD_BENCH_JSON) with its MFMA-roofline fraction.
"""
import argparse
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, emit_json, package, sync  # noqa: E402


def get_args():
    p = argparse.ArgumentParser(description="hgemm benchmark")
    for flag in ("--M", "--N", "--K", "--MNK"):
        p.add_argument(flag, type=int, default=None)
    p.add_argument("--MMNK", type=int, default=12800, help="Matrix MAX M=M=N=K size")
    p.add_argument("--SEP", "--sep", type=int, default=256, help="Matrix SEP M=M=N=K size")
    p.add_argument("--warmup", "--w", type=int, default=2)
    p.add_argument("--iters", "--i", type=int, default=10)
    p.add_argument("--verbose", "--v", action="store_true")
    p.add_argument("--show-matrix", "--show-m", action="store_true")
    p.add_argument("--show-all-info", "--show-a", action="store_true")
    p.add_argument("--show-memory", "--show-mm", action="store_true")
    p.add_argument("--enable-mma", "--mma", action="store_true")
    p.add_argument("--enable-mma-tn", "--mma-tn", action="store_true")
    p.add_argument("--enable-wmma", "--wmma", action="store_true")
    p.add_argument("--enable-cuda", "--cuda", action="store_true")
    p.add_argument("--enable-mma-all", "--mma-all", action="store_true")
    p.add_argument("--enable-wmma-all", "--wmma-all", action="store_true")
    p.add_argument("--enable-cuda-all", "--cuda-all", action="store_true")
    p.add_argument("--enable-torch", "--torch", action="store_true")
    p.add_argument("--enable-cute-tn", "--cute-tn", action="store_true")
    p.add_argument("--enable-cute", "--cute", action="store_true")
    p.add_argument("--disable-cublas", "--no-cublas", action="store_true")
    p.add_argument("--disable-cublas-tn", "--no-cublas-tn", action="store_true")
    p.add_argument("--sleep-duration", "--sleep", type=float, default=0.1)
    p.add_argument("--swizzle-factor", "--swizzle", type=float, default=None)
    p.add_argument("--no-default", action="store_true")
    p.add_argument("--plot-flops", "--plot", action="store_true")
    p.add_argument("--plot-topk", "--topk", type=int, default=8)
    p.add_argument("--no-plot-best", "--no-best", action="store_true")
    p.add_argument("--exclude-tags", "--exclude", type=str, default=None)
    p.add_argument("--save-dir", "--dir", type=str, default="./")
    p.add_argument("--save-tag", "--tag", type=str, default=None)
    p.add_argument("--force-build", "--build", action="store_true")
    return p.parse_args()


W4 = "hgemm_mma_m16n8k16_mma2x4_warp4x4"
W4X2 = W4 + "x2_stages_dsmem"


def row_table(a):
    """(enabled, tag, function, stages or None, block swizzle, TN operand). Order = reference hgemm.py:320-421."""
    default = not a.no_default
    rows = []
    add = lambda en, tag, fn, st=None, swz=False, tn=False: rows.append((en, tag, fn, st, swz, tn))
    add(a.enable_cuda_all, "(naive)", "hgemm_naive_f16")
    add(a.enable_cuda_all, "(f16x8pack+t8x8+bcf)", "hgemm_t_8x8_sliced_k_f16x8_pack_bcf")
    cu = a.enable_cuda or a.enable_cuda_all
    add(cu, "(f16x8pack+t8x8+dbuf)", "hgemm_t_8x8_sliced_k_f16x8_pack_bcf_dbuf")
    add(cu, "(f16x8pack+t8x8+k16+dbuf)", "hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf")
    wm, wma = (a.enable_wmma or a.enable_wmma_all), a.enable_wmma_all
    add(wm, "(wmma4x2)", "hgemm_wmma_m16n16k16_mma4x2")
    add(wm, "(wmma4x2+warp2x4)", "hgemm_wmma_m16n16k16_mma4x2_warp2x4")
    f = "hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem"
    for st, sw in ((3, False), (2, False), (3, True), (2, True)):
        add(wm, "(wmma4x2+warp2x4+stage%d+dsmem%s)" % (st, "+swizzle<block>" if sw else ""), f, st, sw)
    f = "hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages"
    for st, sw in ((3, False), (2, False), (3, True), (2, True)):
        add(wma, "(wmma4x2+warp2x4+stage%d%s)" % (st, "+swizzle<block>" if sw else ""), f, st, sw)
    for sw in (False, True):
        for nm, f in (("wmma4x4+warp4x4", "hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem"),
                      ("wmma4x2+warp4x4", "hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem")):
            for st in (3, 2):
                add(wma, "(%s+stage%d+dsmem%s)" % (nm, st, "+swizzle<block>" if sw else ""), f, st, sw)
    mm, mma = (a.enable_mma or a.enable_mma_all), a.enable_mma_all
    add(mma, "(mma2x4+warp4x4)", W4)
    for st in (3, 2):
        add(mma, "(mma2x4+warp4x4+stage%d)" % st, W4 + "_stages", st)
    for st in (3, 2):
        add(mm, "(mma2x4+warp4x4+stage%d+dsmem)" % st, W4 + "_stages_dsmem", st)
    for st in (4, 3, 2):
        add(mm, "(mma2x4+warp4x4x2+stage%d+dsmem)" % st, W4X2, st)
    for st in (4, 3, 2):
        add(mm, "(mma2x4+warp4x4x2+stage%d+dsmem+swizzle<smem>)" % st, W4X2 + "_swizzle", st)
    for suffix in ("rr", "x4"):
        for st in (4, 3, 2):
            add(mma, "(mma2x4+warp4x4x2+stage%d+dsmem+%s)" % (st, suffix), W4X2 + "_" + suffix, st)
    for st in (3, 2):
        add(mm, "(mma2x4+warp4x4+stage%d+dsmem+swizzle<block>)" % st, W4 + "_stages_dsmem", st, True)
    for st in (4, 3, 2):
        add(mm, "(mma2x4+warp4x4x2+stage%d+dsmem+swizzle<block>)" % st, W4X2, st, True)
    for st in (4, 3, 2):
        add(mm, "(mma2x4+warp4x4x2+stage%d+dsmem+swizzle<smem+block>)" % st, W4X2 + "_swizzle", st, True)
    for st in (3, 2):
        add(mma, "(mma2x4+warp4x4+stage%d+swizzle<block>)" % st, W4 + "_stages", st, True)
    for suffix in ("rr", "x4"):
        for st in (4, 3, 2):
            add(mma, "(mma2x4+warp4x4x2+stage%d+dsmem+swizzle<block>+%s)" % (st, suffix), W4X2 + "_" + suffix, st, True)
    tn = a.enable_mma_tn
    for sw in (False, True):
        sfx = "+swizzle<block>" if sw else ""
        for st in (3, 2):
            add(tn, "tn(mma2x4+warp4x4+stage%d+dsmem%s)" % (st, sfx), W4 + "_stages_dsmem_tn", st, sw, True)
        for st in (4, 3, 2):
            add(tn, "tn(mma2x4+warp4x4x2+stage%d+dsmem+swizzle<smem%s>)" % (st, "+block" if sw else ""),
                W4X2 + "_tn_swizzle_x4", st, sw, True)
    ct = a.enable_cute_tn or a.enable_cute
    for sw in (False, True):
        for st in (4, 3, 2):
            add(ct, "tn(cute+stage%d+swizzle<smem%s>)" % (st, "+block" if sw else ""),
                "hgemm_mma_stages_block_swizzle_tn_cute", st, sw, True)
    add(not a.disable_cublas_tn and (tn or ct), "tn(cublas)", "hgemm_cublas_tensor_op_tn", None, False, True)
    add(not a.disable_cublas and default, "(cublas)", "hgemm_cublas_tensor_op_nn")
    return [r for r in rows if r[0]]


def main():
    args = get_args()
    from cuda_learn_notes_amd import bench_utils as bu
    bu.pretty_print_line()
    print(args)
    bu.pretty_print_line()
    if not HAS_GPU:
        print("no GPU: only the torch.matmul row can run (CPU); kernel rows need the HIP library")
    hgemm = package().hgemm_lib() if HAS_GPU else None
    state = {"max": -1.0}
    json_rows = []

    def bench(tag, call, M, N, K, out, swizzle_stride, is_cublas=False):
        if is_cublas:
            hgemm.init_cublas_handle()
        out.fill_(0)
        for _ in range(args.warmup):
            call()
        sync()
        t0 = time.time()
        for _ in range(args.iters):
            call()
        sync()
        secs = (time.time() - t0) / args.iters
        flat = out.flatten()
        vals = [f"{round(v, 8):<12}"[:10] for v in (flat[0].item(), flat[-1].item())]
        tflops = 2.0 * M * N * K * 1e-12 / secs
        ms = str(f"{secs * 1000:<12}")[:8]
        stride_txt = "NOOP" if swizzle_stride == 1 else swizzle_stride
        line = f"{tag:>53}: {vals}, time:{ms}ms, swizzle<block>: {stride_txt:<4}, TFLOPS: {tflops:<6.2f}"
        if tflops > state["max"]:
            imp = 0 if state["max"] <= 0 else round((tflops - state["max"]) / state["max"] * 100, 2)
            state["max"] = tflops
            print(line + f"(+{imp:.2f}%)")
        elif args.show_all_info or is_cublas:
            print(line)
        if args.show_matrix:
            print(out)
        json_rows.append({"kernel": tag, "shape": [M, N, K], "ms": secs * 1e3, "tflops": tflops,
                          "roofline": {"bound": "mfma", "peak": bu.PEAK_FP16_MFMA_TFLOPS, "achieved": tflops,
                                       "frac": tflops / bu.PEAK_FP16_MFMA_TFLOPS}})
        if is_cublas:
            hgemm.destroy_cublas_handle()
        gc.collect()
        time.sleep(args.sleep_duration if HAS_GPU else 0)

    if args.MNK:
        Ms = Ns = Ks = [args.MNK]
    elif args.M and args.N and args.K:
        Ms, Ns, Ks = [args.M], [args.N], [args.K]
    else:
        Ms = Ns = Ks = list(range(args.SEP, args.MMNK + args.SEP, args.SEP))
    MAX_M, MAX_N, MAX_K = max(Ms), max(Ns), max(Ks)
    torch.manual_seed(int(os.environ.get("CLN_AMD_SEED", "0")))
    A = torch.randn((MAX_M, MAX_K), dtype=torch.half, device=DEVICE)
    B = torch.randn((MAX_K, MAX_N), dtype=torch.half, device=DEVICE)
    C = torch.randn((MAX_M, MAX_N), dtype=torch.half, device=DEVICE)
    rows = row_table(args) if HAS_GPU else []
    for M, N, K in zip(Ms, Ns, Ks):
        state["max"] = -1.0
        bu.pretty_print_line()
        bu.pretty_print_line(f"M={M}, N={N}, K={K}, Warmup={args.warmup}, Iters={args.iters}, {len(rows) + 1}/{len(rows) + 1}", " ")
        bu.pretty_print_line()
        a = A[:M, :K].contiguous()
        b = B[:K, :N].contiguous()
        c = C[:M, :N].contiguous()
        b_col_major = bu.as_col_major(b) if any(r[5] for r in rows) else None
        for _, tag, fname, stages, swz, tn in rows:
            fn = getattr(hgemm, fname)
            bb = b_col_major if tn else b
            stride = bu.make_block_swizzle_stride(N, K, args.swizzle_factor) if swz else 1
            swz_on = swz and stride >= 256
            if stages is None:
                call = lambda fn=fn, bb=bb: fn(a, bb, c)
            else:
                call = lambda fn=fn, bb=bb, st=stages, so=swz_on, sd=stride: fn(a, bb, c, st, so, sd)
            try:
                bench(tag, call, M, N, K, c, stride, "cublas" in tag)
            except RuntimeError as e:
                print(f"{tag:>53}: skipped ({e})")
        if args.enable_torch or not HAS_GPU:
            bench("(torch)", lambda: torch.matmul(a, b, out=c), M, N, K, c, 1)
        sync()
        bu.pretty_print_line()
    if args.plot_flops:
        print("--plot-flops: matplotlib is not available in this image; per-row numbers are in $CLN_AMD_BENCH_JSON")
    if args.show_memory and HAS_GPU:
        bu.pretty_print_line()
        print(torch.cuda.memory_summary())
    emit_json(json_rows)


if __name__ == "__main__":
    main()
