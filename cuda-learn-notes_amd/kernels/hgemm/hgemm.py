"""HGEMM bench driver -- same CLI flags, row tags, printed columns AND helper functions as reference
kernels/hgemm/hgemm.py: get_args :16-52, make_block_swizzle_stride :71-81, run_benchmark :84-192 (same parameters and
return value), get_topk_tflops :195-208, get_best_tflops :211-220, plot_tflops :223-274, get_mnk :277-281, tag table
:320-421 (re-authored as a table walked by one loop). `from hgemm import run_benchmark` works without a GPU; the kernel
rows need one (no CPU fallback).

  python hgemm.py --mma --MNK 4096          # BASELINE config C3
  python hgemm.py --mma-all --wmma-all --cuda-all --mma-tn --cute-tn --torch

Rows whose tag contains "cublas" run the rocBLAS row, "(hipblaslt)" / "tn(hipblaslt)" the hipBLASLt row (libcln_amd_vendor.so). Each row also goes to the
JSON side channel ($CLN_AMD_BENCH_JSON) with its MFMA-roofline fraction.
"""
import argparse
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, emit_json, sync  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tools.utils import (as_col_major, get_device_name, pretty_print_line,  # noqa: E402  (reference hgemm.py:8-11)
                         try_load_hgemm_library)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_learn_notes_amd import bench_utils as bu  # noqa: E402


def get_args(argv=None):
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
    return p.parse_args(argv)


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
    # the second vendor baseline (not in the reference: hipBLASLt is what MI355X GEMM numbers are usually quoted on)
    add(not a.disable_cublas_tn and (tn or ct), "tn(hipblaslt)", "cln_hgemm_hipblaslt_tn", None, False, True)
    add(not a.disable_cublas and default, "(hipblaslt)", "cln_hgemm_hipblaslt_nn")
    return [r for r in rows if r[0]]


# the reference parses its flags at import (hgemm.py:55); imported as a module (tests) this uses the defaults
args = get_args(None if __name__ == "__main__" else [])
hgemm = None  # set by main() / load_library(): the module object with the 38 exported functions

MAX_TFLOPS = -1
STATIS_INFO = {}
TOATL_TFLOPS = {}  # (sic) reference spelling, hgemm.py:66
CUBLAS_TOTAL_TFLOPS = 0
CUBLAS_TN_TOTAL_TFLOPS = 0
JSON_ROWS = []

make_block_swizzle_stride = bu.make_block_swizzle_stride  # reference hgemm.py:71-81


def load_library(force_build: bool = False, verbose: bool = False):
    global hgemm
    hgemm = try_load_hgemm_library(force_build=force_build, verbose=verbose)
    return hgemm


@torch.no_grad()
def run_benchmark(perf_func, a, b, tag, out=None, stages=-1, swizzle=False, swizzle_stride=1, warmup=None, iters=None,
                  show_matrix=None, only_show_improved=None):
    """Time one HGEMM row with the reference protocol (warmup, synchronize, time.time() around `iters` async launches,
    synchronize) and print it in the reference format. `perf_func(a, b, out[, stages, swizzle, swizzle_stride])`;
    `b` keeps the [K, N] shape for the TN rows. Returns (out, mean_time_ms) like reference hgemm.py:84-192."""
    global MAX_TFLOPS, CUBLAS_TOTAL_TFLOPS, CUBLAS_TN_TOTAL_TFLOPS
    warmup = args.warmup if warmup is None else warmup
    iters = args.iters if iters is None else iters
    show_matrix = args.show_matrix if show_matrix is None else show_matrix
    only_show_improved = (not args.show_all_info) if only_show_improved is None else only_show_improved
    M, K, N = a.size(0), a.size(1), b.size(1)
    if swizzle:
        swizzle_stride = make_block_swizzle_stride(N, K, args.swizzle_factor)
        swizzle = swizzle if swizzle_stride >= 256 else False
    else:
        swizzle_stride = 1  # no thread-block swizzle
    is_cublas = "cublas" in tag
    if is_cublas and hgemm is not None:
        hgemm.init_cublas_handle()
    if out is not None:
        out.fill_(0)

    def call():
        if out is None:
            return perf_func(a, b)
        if stages > 1:
            perf_func(a, b, out, stages, swizzle, swizzle_stride)
        else:
            perf_func(a, b, out)
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
    out = res
    flat = out.flatten()
    out_val = [f"{round(v, 8):<12}"[:10] for v in (flat[0].item(), flat[-1].item())]
    TFLOPS = (2 * M * N * K) * 1e-12 / mean_secs
    mean_time_ms = str(f"{mean_secs * 1000:<12}")[:8]
    stride_txt = "NOOP" if swizzle_stride == 1 else swizzle_stride
    line = f"{tag:>53}: {out_val}, time:{mean_time_ms}ms, swizzle<block>: {stride_txt:<4}, TFLOPS: {TFLOPS:<6.2f}"
    if TFLOPS > MAX_TFLOPS:
        improve = round((TFLOPS - MAX_TFLOPS) / MAX_TFLOPS * 100, 2) if MAX_TFLOPS > 0 else 0
        MAX_TFLOPS = TFLOPS
        print(line + f"(+{improve:.2f}%)")
    elif not only_show_improved or is_cublas or "hipblaslt" in tag:
        print(line)
    if show_matrix:
        print(out)
    if args.plot_flops:
        STATIS_INFO.setdefault(tag, []).append(TFLOPS)
        if "hipblaslt" in tag:
            pass  # a baseline row: not a candidate for the top-k table
        elif not is_cublas:
            TOATL_TFLOPS[tag] = TOATL_TFLOPS.get(tag, 0) + TFLOPS
        elif tag == "tn(cublas)":
            CUBLAS_TN_TOTAL_TFLOPS += TFLOPS
        else:
            CUBLAS_TOTAL_TFLOPS += TFLOPS
    JSON_ROWS.append({"kernel": tag, "shape": [M, N, K], "ms": mean_secs * 1e3, "tflops": TFLOPS,
                      "roofline": {"bound": "mfma", "peak": bu.PEAK_FP16_MFMA_TFLOPS, "achieved": TFLOPS,
                                   "frac": TFLOPS / bu.PEAK_FP16_MFMA_TFLOPS}})
    sync()
    if is_cublas and hgemm is not None:
        hgemm.destroy_cublas_handle()
    gc.collect()
    time.sleep(args.sleep_duration if HAS_GPU else 0)
    return out, mean_time_ms


def get_topk_tflops():
    """Print the per-algorithm TFLOPS totals over the size sweep and return the top-k tags (reference :195-208)."""
    topk = sorted(TOATL_TFLOPS.items(), key=lambda kv: kv[1], reverse=True)
    pretty_print_line()
    pretty_print_line(f"THE TOTAL TFLOPS OF {len(topk)} HGEMM ALGO ON {get_device_name()} DEVICE", " ")
    pretty_print_line()
    for tag, tflops in topk[::-1]:
        print(f"{tag:>53}: {tflops:>20.2f} TFLOPS")
    if CUBLAS_TN_TOTAL_TFLOPS > 1:
        print(f"{'tn(cublas)':>53}: {CUBLAS_TN_TOTAL_TFLOPS:>20.2f} TFLOPS")
    if CUBLAS_TOTAL_TFLOPS > 1:
        print(f"{'(cublas)':>53}: {CUBLAS_TOTAL_TFLOPS:>20.2f} TFLOPS")
    pretty_print_line()
    return [t for t, _ in topk[:args.plot_topk]]


def get_best_tflops():
    """Per size, the best TFLOPS over all non-vendor rows (reference :211-220)."""
    rows = [v for t, v in STATIS_INFO.items() if "cublas" not in t and "hipblaslt" not in t and "MNK" not in t and t != "(best)"]
    n = min(len(r) for r in rows) if rows else 0
    return [max(r[i] for r in rows) for i in range(n)]


def plot_tflops():
    """TFLOPS-vs-size chart of the top-k rows, the vendor rows and the per-size best (reference :223-274). Written as
    <save_dir>/<device>[_<tag>].png with matplotlib when it is importable, else as .svg by kernels/_svgplot.py."""
    exclude = set((args.exclude_tags.split(",") if args.exclude_tags else []) + ["MNK"])
    draw = get_topk_tflops() + ["(cublas)", "tn(cublas)"]
    STATIS_INFO["(best)"] = get_best_tflops()
    draw.append("(best)")
    xs = STATIS_INFO.get("MNK", [])
    series = []
    for tag, tfl in STATIS_INFO.items():
        if tag not in draw or any(e in tag for e in exclude) or not tfl:
            continue
        if "best" in tag and args.no_plot_best:
            continue
        style = "bold" if tag in ("(cublas)", "tn(cublas)", "(best)") else "dash"
        series.append((tag, list(tfl), style))
    device_name = get_device_name().replace(" ", "_")
    stem = f"{args.save_dir}/{device_name}_{args.save_tag}" if args.save_tag else f"{args.save_dir}/{device_name}"
    os.makedirs(args.save_dir, exist_ok=True)
    title = f"My HGEMM vs rocBLAS, {get_device_name()}, Warmup={args.warmup}, Iters={args.iters}"
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        ax = plt.subplots(figsize=(16, 9))[1]
        ax.set_title(title), ax.set_xlabel("M=N=K"), ax.set_ylabel("TFLOPS"), ax.grid(True)
        ax.set_xticks(range(len(xs))), ax.set_xticklabels(xs, rotation=45, ha="right")
        for tag, tfl, style in series:
            ax.plot(tfl, label=tag, linewidth=3 if style == "bold" else 1.5, linestyle="-" if style == "bold" else "--")
        ax.legend()
        save_path = stem + ".png"
        plt.savefig(save_path, dpi=300)
    except ImportError:
        from _svgplot import line_chart
        save_path = line_chart(stem + ".svg", title, [str(x) for x in xs], series)
    pretty_print_line(f"plot hgemm TFLOPS done, saved as {save_path}")
    return save_path


def get_mnk(sep: int = None):
    sep = args.SEP if sep is None else sep
    r = list(range(sep, args.MMNK + sep, sep))
    return r, list(r), list(r)


def main():
    global MAX_TFLOPS
    pretty_print_line()
    print(args)
    pretty_print_line()
    if not HAS_GPU:
        print("no GPU: only the torch.matmul row can run (CPU); kernel rows need the HIP library")
    lib = load_library(force_build=args.force_build, verbose=args.verbose) if HAS_GPU else None
    Ms, Ns, Ks = get_mnk()
    if args.MNK:
        Ms = Ns = Ks = [args.MNK]
    if args.M and args.N and args.K:
        Ms, Ns, Ks = [args.M], [args.N], [args.K]
    STATIS_INFO["MNK"] = list(Ms)
    MAX_M, MAX_N, MAX_K = max(Ms), max(Ns), max(Ks)
    torch.manual_seed(int(os.environ.get("CLN_AMD_SEED", "0")))
    sync()
    start = time.time()
    pretty_print_line(f"Allocate buffers for fast profiling start, MAX_M={MAX_M}, MAX_N={MAX_N}, MAX_K={MAX_K}")
    A = torch.randn((MAX_M, MAX_K), dtype=torch.half, device=DEVICE)
    B = torch.randn((MAX_K, MAX_N), dtype=torch.half, device=DEVICE)
    C = torch.randn((MAX_M, MAX_N), dtype=torch.half, device=DEVICE)
    sync()
    pretty_print_line(f"Allocate buffers for fast profiling done, time: {(time.time() - start) * 1000:.7f} ms")
    rows = row_table(args) if HAS_GPU else []
    for count, (M, N, K) in enumerate(zip(Ms, Ns, Ks), 1):
        MAX_TFLOPS = -1
        pretty_print_line()
        pretty_print_line(f"M={M}, N={N}, K={K}, Warmup={args.warmup}, Iters={args.iters}, {count}/{len(Ms)}", " ")
        pretty_print_line()
        a = A[:M, :K].contiguous()
        b = B[:K, :N].contiguous()
        c = C[:M, :N].contiguous()
        b_col_major = as_col_major(b) if any(r[5] for r in rows) else None
        for _, tag, fname, stages, swz, tn in rows:
            if not hasattr(lib, fname):  # an optional comparison row (hipBLASLt) the vendor library was built without
                print(f"{tag:>53}: skipped (row not built)")
                continue
            try:
                run_benchmark(getattr(lib, fname), a, b_col_major if tn else b, tag, c,
                              stages=-1 if stages is None else stages, swizzle=swz)
            except RuntimeError as e:
                print(f"{tag:>53}: skipped ({e})")
        if args.enable_torch or not HAS_GPU:
            run_benchmark(lambda x, y, out: torch.matmul(x, y, out=out), a, b, "(torch)", c)
        sync()
        pretty_print_line()
    if args.plot_flops:
        plot_tflops()
    if args.show_memory and HAS_GPU:
        pretty_print_line()
        print(torch.cuda.memory_summary())
    emit_json(JSON_ROWS)


if __name__ == "__main__":
    main()
