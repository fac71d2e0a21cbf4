"""sgemm bench -- rows/tags of reference kernels/sgemm/sgemm.py:120-152 (fp32 CUDA-core rungs, cuBLAS, TF32 WMMA
rungs -> here exact-f32 MFMA). `--MNK n` limits the size sweep (reference sweeps 4096..16384; default here
4096 and 8192). No GPU: only the torch.matmul row runs, on CPU, at 1024^3."""
import argparse
import os
import sys
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, package, timed  # noqa: E402

lib = package().load("sgemm", "sgemm_vendor") if HAS_GPU else None


def row(tag, call, c, M, N, K, warmup, iters):
    if call is None:
        print(f"{'out_' + tag:>42}: skipped (no GPU: the HIP kernel path has no CPU fallback)")
        return
    _, ms = timed(call, warmup, iters)
    vals = [round(v, 6) for v in c.flatten()[:2].tolist()]
    print(f"{'out_' + tag:>42}: {vals}, time:{ms:<.6f}ms, TFLOPS:{2.0 * M * N * K / ms * 1e-9:<6.2f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--MNK", type=int, default=None)
    args = ap.parse_args()
    sizes = [args.MNK] if args.MNK else ([4096, 8192] if HAS_GPU else [1024])
    warmup, iters = (2, 10) if HAS_GPU else (0, 1)
    for S in sizes:
        M = N = K = S
        print("-" * 130)
        print(" " * 55 + f"M={M}, N={N}, K={K}")
        a = torch.randn((M, K)).to(DEVICE).float().contiguous()
        b = torch.randn((K, N)).to(DEVICE).float().contiguous()
        c = torch.zeros((M, N)).to(DEVICE).float().contiguous()
        k3 = lambda n: None if lib is None else partial(getattr(lib, n), a, b, c)
        k6 = lambda n, st, sw: None if lib is None else partial(getattr(lib, n), a, b, c, st, sw, 2048)
        row("f32x4(t8x8sk)", k3("sgemm_t_8x8_sliced_k_f32x4"), c, M, N, K, warmup, iters)
        row("f32x4(t8x8bcf)", k3("sgemm_t_8x8_sliced_k_f32x4_bcf"), c, M, N, K, warmup, iters)
        row("f32x4(t8x8dbuf)", k3("sgemm_t_8x8_sliced_k_f32x4_bcf_dbuf"), c, M, N, K, warmup, iters)
        row("f32x4(t8x8k16dbuf+async)", k3("sgemm_t_8x8_sliced_k16_f32x4_bcf_dbuf_async"), c, M, N, K, warmup, iters)
        row("f32x4(t8x16k16dbuf+async)", k3("sgemm_t_8x16_sliced_k16_f32x4_bcf_dbuf_async"), c, M, N, K, warmup, iters)
        row("f32(cublas)", k3("sgemm_cublas"), c, M, N, K, warmup, iters)
        print("-" * 62 + "WMMA" + "-" * 64)
        for st in (3, 2):
            for sw in (False, True):
                row(f"tf32(mma2x4+warp2x4+stage{st}{'+swizzle' if sw else ''})",
                    k6("sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages", st, sw), c, M, N, K, warmup, iters)
        row("tf32(cublas+tf32)", k3("sgemm_cublas_tf32"), c, M, N, K, warmup, iters)
        row("f32_th", partial(torch.matmul, a, b, out=c), c, M, N, K, warmup, iters)
    print("-" * 130)


if __name__ == "__main__":
    main()
