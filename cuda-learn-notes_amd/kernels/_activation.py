"""Shared driver of the seven activation bench scripts (reference kernels/{relu,sigmoid,gelu,swish,elu,hardswish,
hardshrink}/<op>.py share one layout: S,K in {1024,2048,4096}^2, six kernel rows + the torch row per dtype)."""
import os
import sys
from functools import partial

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import DEVICE, HAS_GPU, emit_json, package, run_table  # noqa: E402

TORCH = {
    "relu": torch.relu, "sigmoid": torch.sigmoid, "gelu": partial(F.gelu, approximate="tanh"),
    "swish": lambda x: x * torch.sigmoid(x), "elu": F.elu, "hardswish": F.hardswish,
    "hardshrink": partial(F.hardshrink, lambd=0.5),
}


def main(op):
    lib = package().load("activation") if HAS_GPU else None
    warmup, iters = (10, 1000) if HAS_GPU else (1, 3)
    sizes = [1024, 2048, 4096] if HAS_GPU else [1024]
    json_rows, sections = [], []

    def k(name, x, y):
        return None if lib is None else partial(getattr(lib, name), x, y)

    for S in sizes:
        for K in sizes:
            x = torch.randn((S, K)).to(DEVICE).float().contiguous()
            y = torch.zeros_like(x)
            nb = 2 * x.numel() * 4
            rows = [(r, k("%s_%s" % (op, r), x, y), y, x.shape, nb) for r in ("f32", "f32x4")]
            rows.append(("f32_th", partial(TORCH[op], x), None, x.shape, nb))
            sections.append((f"S={S}, K={K}", rows, warmup, iters))
            xh, yh = x.half().contiguous(), y.half().contiguous()
            rows = [(r.replace("_pack", "pack"), k("%s_%s" % (op, r), xh, yh), yh, xh.shape, nb // 2)
                    for r in ("f16", "f16x2", "f16x8", "f16x8_pack")]
            rows.append(("f16_th", partial(TORCH[op], xh), None, xh.shape, nb // 2))
            sections.append((f"S={S}, K={K} (f16)", rows, warmup, iters))
    run_table(85, sections, out_width=18, json_rows=json_rows)
    emit_json(json_rows)
