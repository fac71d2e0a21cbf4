"""elementwise add bench -- same rows/columns as reference kernels/elementwise/elementwise.py:59-82.
No GPU: only the torch rows run, on CPU (BASELINE config C1: elementwise_add_f32, N = 4 Mi)."""
import os
import sys
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, emit_json, hbm_row, package, timed  # noqa: E402

lib = package().load("elementwise") if HAS_GPU else None

F32_ROWS = [("f32", "elementwise_add_f32"), ("f32x4", "elementwise_add_f32x4")]
F16_ROWS = [("f16", "elementwise_add_f16"), ("f16x2", "elementwise_add_f16x2"), ("f16x8", "elementwise_add_f16x8"),
            ("f16x8pack", "elementwise_add_f16x8_pack")]


def show(tag, out, ms):
    vals = [round(v, 8) for v in out.flatten()[:2].tolist()]
    print(f"{'out_' + tag:>18}: {vals}, time:{ms:.8f}ms")


def run_rows(rows, a, b, c, th_tag, warmup, iters, json_rows):
    nbytes = 3 * a.numel() * a.element_size()
    for tag, name in rows:
        if lib is None:
            print(f"{'out_' + tag:>18}: skipped (no GPU: the HIP kernel path has no CPU fallback)")
            continue
        c.fill_(0)
        fn = getattr(lib, name)
        _, ms = timed(lambda: fn(a, b, c), warmup, iters)
        show(tag, c, ms)
        json_rows.append(hbm_row(name, a.shape, ms, nbytes))
    out, ms = timed(partial(torch.add, a, b, out=c), warmup, iters)
    show(th_tag, c, ms)
    json_rows.append(hbm_row("torch.add(%s,%s)" % (a.dtype, DEVICE.type), a.shape, ms, nbytes))


def main():
    warmup, iters = (10, 1000) if HAS_GPU else (2, 20)
    Ss = Ks = [1024, 2048, 4096]
    json_rows = []
    for S in Ss:
        for K in Ks:
            print("-" * 85)
            print(" " * 40 + f"S={S}, K={K}")
            a = torch.randn((S, K)).to(DEVICE).float().contiguous()
            b = torch.randn((S, K)).to(DEVICE).float().contiguous()
            c = torch.zeros_like(a)
            run_rows(F32_ROWS, a, b, c, "f32_th", warmup, iters, json_rows)
            print("-" * 85)
            ah, bh, ch = a.half().contiguous(), b.half().contiguous(), c.half().contiguous()
            run_rows(F16_ROWS, ah, bh, ch, "f16_th", warmup, iters, json_rows)
            print("-" * 85)
    emit_json(json_rows)


if __name__ == "__main__":
    main()
