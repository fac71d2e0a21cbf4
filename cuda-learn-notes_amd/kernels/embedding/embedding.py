"""embedding bench -- same rows/tags/shapes as reference kernels/embedding/embedding.py:72-98.
No GPU: only the torch.nn.functional.embedding rows run, on CPU."""
import os
import sys
from functools import partial

import torch
from torch.nn.functional import embedding

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, emit_json, package, run_table  # noqa: E402

lib = package().load("embedding") if HAS_GPU else None


def k(name, i, w, o):
    return None if lib is None else partial(getattr(lib, name), i, w, o)


def main():
    json_rows, sections = [], []
    for M in (1024, 4096):
        for N in (2048, 4096):
            for K in (512, 1024):
                i = torch.randint(0, M, size=(N,)).to(DEVICE).int().contiguous()
                w = torch.randn((M, K)).float().to(DEVICE).contiguous()
                o = torch.zeros((N, K)).float().to(DEVICE).contiguous()
                nb = 2 * o.numel() * 4 + N * 4
                il = i.long()
                rows = [(t, k(n, i, w, o), o, o.shape, nb) for t, n in
                        (("f32", "embedding_f32"), ("f32x4", "embedding_f32x4"), ("f32x4_pack", "embedding_f32x4_pack"))]
                rows.append(("f32_th", partial(embedding, il, w), None, o.shape, nb))
                sections.append((f"MaxV={M}, SeqLen={N}, EmbSize={K}", rows, 2, 20))
                wh, oh = w.half(), o.half()
                rows = [(t, k(n, i, wh, oh), oh, oh.shape, nb // 2) for t, n in
                        (("f16", "embedding_f16"), ("f16x8", "embedding_f16x8"), ("f16x8_pack", "embedding_f16x8_pack"))]
                rows.append(("f16_th", partial(embedding, il, wh), None, oh.shape, nb // 2))
                sections.append((f"MaxV={M}, SeqLen={N}, EmbSize={K} (f16)", rows, 2, 20))
    run_table(110, sections, out_width=23, json_rows=json_rows)
    emit_json(json_rows)


if __name__ == "__main__":
    main()
