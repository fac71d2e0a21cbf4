"""hgemv bench -- same rows/tags as reference kernels/hgemv/hgemv.py:61-76 (M=1024: K=128 then K=16).
No GPU: only the torch.matmul rows run, on CPU."""
import os
import sys
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, package, run_table  # noqa: E402

lib = package().load("hgemv") if HAS_GPU else None


def k(name, a, b, c):
    return None if lib is None else partial(getattr(lib, name), a, b, c)


def main():
    sections = []
    for M, K, rows in ((1024, 128, (("k32f16", "hgemv_k32_f16"), ("k128f16x4", "hgemv_k128_f16x4"))),
                       (1024, 16, (("k16f16", "hgemv_k16_f16"),))):
        a = torch.randn((M, K)).to(DEVICE).half().contiguous()
        b = torch.randn((K, 1)).to(DEVICE).half().contiguous()
        c = torch.zeros((M, 1)).to(DEVICE).half().contiguous()
        nb = a.numel() * a.element_size()
        rs = [(t, k(n, a, b, c), c, a.shape, nb) for t, n in rows]
        rs.append(("f16_th", partial(torch.matmul, a, b, out=c), c, a.shape, nb))
        sections.append((f"M={M}, N=1, K={K}", rs, 10, 1000 if HAS_GPU else 5))
    run_table(80, sections, out_width=14)


if __name__ == "__main__":
    main()
