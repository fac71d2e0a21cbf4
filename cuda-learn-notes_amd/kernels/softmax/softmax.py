"""softmax bench -- same rows/tags as reference kernels/softmax/softmax.py:58-230 (global "fence" softmax on
N = 128*128, then per-token rows for S=4096, H in {256..8192}). No GPU: only the torch rows run, on CPU."""
import os
import sys
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, emit_json, package, run_table  # noqa: E402

lib = package().load("softmax") if HAS_GPU else None


def k(name, *args):
    return None if lib is None else partial(getattr(lib, name), *args)


def main():
    warmup, iters = (10, 1000) if HAS_GPU else (1, 5)
    json_rows, sections = [], []
    N = 128 * 128
    x = torch.randn(N, device=DEVICE).float()
    out = torch.zeros_like(x)
    nb = 2 * x.numel() * 4
    sections.append((f"N={N}", [
        ("f32(fence)", k("softmax_f32", x, out), out, x.shape, nb),
        ("f32x4(fence)", k("softmax_f32x4", x, out), out, x.shape, nb),
        ("f32_th", partial(torch.softmax, x, dim=0, out=out), out, x.shape, nb)], warmup, iters))
    for H in (256, 512, 1024, 2048, 4096, 8192):
        S = 4096
        x = torch.randn((S, H), device=DEVICE).float().contiguous()
        out = torch.zeros_like(x)
        nb = 2 * x.numel() * 4
        f32 = [("f32(per)", "softmax_f32_per_token"), ("f32x4(per)", "softmax_f32x4_per_token"),
               ("f32(safe)", "safe_softmax_f32_per_token"), ("f32(safe+online)", "online_safe_softmax_f32_per_token"),
               ("f32x4(safe+online)", "online_safe_softmax_f32x4_pack_per_token"),
               ("f32x4(safe)", "safe_softmax_f32x4_per_token")]
        if H > 1024:  # reference drops the scalar rungs above 1024 threads per row (softmax.py:150-230)
            f32 = [r for r in f32 if "x4" in r[0]]
        rows = [(t, k(n, x, out), out, x.shape, nb) for t, n in f32]
        rows.append(("f32_th(per)", partial(torch.softmax, x, dim=1, out=out), out, x.shape, nb))
        sections.append((f"S={S}, H={H}", rows, warmup, iters))
        xh, oh = x.half().contiguous(), out.half().contiguous()
        nbh = 2 * xh.numel() * 2
        f16 = [("f16f32(safe)", "safe_softmax_f16_f32_per_token"), ("f16x2f32(safe)", "safe_softmax_f16x2_f32_per_token"),
               ("f16x8packf32(safe)", "safe_softmax_f16x8_pack_f32_per_token")]
        if H > 1024:
            f16 = f16[1:] if H <= 2048 else f16[2:]
        rows = [(t, k(n, xh, oh), oh, xh.shape, nbh) for t, n in f16]
        rows.append(("f16_th(per)", partial(torch.softmax, xh, dim=1, out=oh), oh, xh.shape, nbh))
        sections.append((f"S={S}, H={H} (f16)", rows, warmup, iters))
    run_table(100, sections, out_width=24, json_rows=json_rows)
    emit_json(json_rows)


if __name__ == "__main__":
    main()
