"""dot-product bench -- same rows/tags as reference kernels/dot-product/dot_product.py:49-70.
No GPU: only the torch.dot rows run, on CPU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, emit_json, hbm_row, package, timed  # noqa: E402

lib = package().load("dot_product") if HAS_GPU else None


def show(tag, out, ms):
    print(f"{'out_' + tag:>18}: {float(out):<15.8f}, time:{ms:.8f}ms")


def main():
    warmup, iters = (10, 1000) if HAS_GPU else (1, 5)
    sizes = [1024, 2048, 4096]
    json_rows = []
    for S in sizes:
        for K in sizes:
            print("-" * 80)
            print(" " * 40 + f"S={S}, K={K}")
            a = torch.randn((S, K)).to(DEVICE).float()
            b = torch.randn((S, K)).to(DEVICE).float()
            for (x, y, rows, th) in ((a, b, (("f32f32", "dot_prod_f32_f32"), ("f32x4f32", "dot_prod_f32x4_f32")), "f32f32_th"),
                                     (a.half(), b.half(), (("f16f32", "dot_prod_f16_f32"), ("f16x2f32", "dot_prod_f16x2_f32"),
                                                           ("f16x8packf32", "dot_prod_f16x8_pack_f32")), "f16f16_th")):
                nb = 2 * x.numel() * x.element_size()
                for tag, name in rows:
                    if lib is None:
                        print(f"{'out_' + tag:>18}: skipped (no GPU: the HIP kernel path has no CPU fallback)")
                        continue
                    fn = getattr(lib, name)
                    out, ms = timed(lambda: fn(x, y), warmup, iters)
                    show(tag, out.item(), ms)
                    json_rows.append(hbm_row(name, x.shape, ms, nb))
                xf, yf = x.flatten(), y.flatten()
                out, ms = timed(lambda: torch.dot(xf, yf), warmup, iters)
                show(th, out.item(), ms)
                print("-" * 80)
    emit_json(json_rows)


if __name__ == "__main__":
    main()
