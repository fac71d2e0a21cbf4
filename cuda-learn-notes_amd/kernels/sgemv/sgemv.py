"""sgemv bench -- same rows/tags as reference kernels/sgemv/sgemv.py:61-76 (M=1024: K=128 then K=16).
No GPU: only the torch.matmul rows run, on CPU."""
import os
import sys
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, package, run_table  # noqa: E402

lib = package().load("sgemv") if HAS_GPU else None


def k(name, a, b, c):
    return None if lib is None else partial(getattr(lib, name), a, b, c)


def main():
    sections = []
    for M, K, rows in ((1024, 128, (("k32f32", "sgemv_k32_f32"), ("k128f32x4", "sgemv_k128_f32x4"))),
                       (1024, 16, (("k16f32", "sgemv_k16_f32"),))):
        a = torch.randn((M, K)).to(DEVICE).float().contiguous()
        b = torch.randn((K, 1)).to(DEVICE).float().contiguous()
        c = torch.zeros((M, 1)).to(DEVICE).float().contiguous()
        nb = a.numel() * a.element_size()
        rs = [(t, k(n, a, b, c), c, a.shape, nb) for t, n in rows]
        rs.append(("f32_th", partial(torch.matmul, a, b, out=c), c, a.shape, nb))
        sections.append((f"M={M}, N=1, K={K}", rs, 10, 1000 if HAS_GPU else 5))
    run_table(80, sections, out_width=14)


if __name__ == "__main__":
    main()
