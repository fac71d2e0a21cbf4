"""rope bench -- same rows/tags as reference kernels/rope/rope.py:90-104 (M in {4096,8192}, N in {512,1024}).
No GPU: only the naive torch row runs, on CPU."""
import os
import sys
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, emit_json, package, run_table  # noqa: E402

lib = package().load("rope") if HAS_GPU else None


def naive_rope(x, theta=10000.0):
    """The script's check column (reference rope.py:68-88), device-agnostic."""
    dim, seq_len = x.shape[-1], x.shape[-2]
    x_ = torch.view_as_complex(x.float().reshape(*x.shape[:-1], -1, 2))
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
    t = torch.arange(seq_len, device=freqs.device)
    freqs = torch.outer(t, freqs).float().to(x.device)
    freqs_cis = torch.polar(torch.ones_like(freqs), freqs)
    return torch.view_as_real(x_ * freqs_cis).flatten(1).type_as(x)


def k(name, x, out):
    return None if lib is None else partial(getattr(lib, name), x, out)


def main():
    warmup, iters = (2, 20)
    json_rows, sections = [], []
    for M in (4096, 8192):
        for N in (512, 1024):
            x = torch.randn((M, N)).to(DEVICE).float().contiguous()
            out = torch.zeros_like(x)
            nb = 2 * x.numel() * 4
            sections.append((f"M={M}, N={N}", [
                ("f32", k("rope_f32", x, out), out, x.shape, nb),
                ("f32x4_pack", k("rope_f32x4_pack", x, out), out, x.shape, nb),
                ("f32_th", partial(naive_rope, x), None, x.shape, nb)], warmup, iters))
    run_table(100, sections, out_width=20, json_rows=json_rows)
    emit_json(json_rows)


if __name__ == "__main__":
    main()
