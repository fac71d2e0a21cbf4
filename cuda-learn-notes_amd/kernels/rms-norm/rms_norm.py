"""rms-norm bench -- same rows/tags as reference kernels/rms-norm/rms_norm.py:68-172.
No GPU: only the naive torch rows run, on CPU."""
import os
import sys
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, emit_json, package, run_table  # noqa: E402

lib = package().load("rms_norm") if HAS_GPU else None


def naive_rms_norm(x, g):
    """The script's check column (reference rms_norm.py:26-31: no eps)."""
    s_rms = torch.rsqrt(torch.mean(x ** 2, dim=1, keepdim=True))
    return x * s_rms * g


def k(name, x, out):
    return None if lib is None else partial(getattr(lib, name), x, out, 1.0)


def main():
    warmup, iters = (10, 1000) if HAS_GPU else (1, 5)
    json_rows, sections = [], []
    for N, K in ((4096, 512), (4096, 1024), (4096, 2048), (4096, 4096), (4096, 8192), (8192, 8192)):
        x = torch.randn((N, K)).to(DEVICE).float().contiguous()
        out = torch.zeros_like(x)
        nb = 2 * x.numel() * 4
        if K <= 4096:
            rows = [("f32", "rms_norm_f32"), ("f32x4", "rms_norm_f32x4")] if K <= 1024 else [("f32x4", "rms_norm_f32x4")]
            rows = [(t, k(n, x, out), out, x.shape, nb) for t, n in rows]
            rows.append(("f32_th", partial(naive_rms_norm, x, 1.0), None, x.shape, nb))
            sections.append((f"N={N}, K={K}", rows, warmup, iters))
        xh, oh = x.half(), out.half()
        nbh = 2 * xh.numel() * 2
        f16 = [("f16f16", "rms_norm_f16_f16"), ("f16f32", "rms_norm_f16_f32"), ("f16x2f16", "rms_norm_f16x2_f16"),
               ("f16x8f16", "rms_norm_f16x8_f16"), ("f16x8f32", "rms_norm_f16x8_f32"),
               ("f16x8packf16", "rms_norm_f16x8_pack_f16"), ("f16x8packf32", "rms_norm_f16x8_pack_f32")]
        if K > 1024:
            f16 = f16[2:]
        if K > 2048:
            f16 = f16[1:]
        rows = [(t, k(n, xh, oh), oh, xh.shape, nbh) for t, n in f16]
        rows.append(("f16_th", partial(naive_rms_norm, xh, 1.0), None, xh.shape, nbh))
        sections.append((f"N={N}, K={K} (f16)", rows, warmup, iters))
    run_table(85, sections, out_width=17, json_rows=json_rows)
    emit_json(json_rows)


if __name__ == "__main__":
    main()
