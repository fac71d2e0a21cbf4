"""block_all_reduce bench -- same rows/tags as reference kernels/reduce/block_all_reduce.py:46-105.
No GPU: only the torch.sum rows run, on CPU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, emit_json, hbm_row, package, timed  # noqa: E402

lib = package().load("reduce") if HAS_GPU else None

GROUPS = [  # (cast, [(tag, function)], torch tag, torch-input cast)
    (lambda v: v, [("f32f32", "block_all_reduce_sum_f32_f32"), ("f32x4f32", "block_all_reduce_sum_f32x4_f32")],
     "f32f32_th", None),
    (lambda v: v.half(), [("f16f16", "block_all_reduce_sum_f16_f16"), ("f16f32", "block_all_reduce_sum_f16_f32"),
                          ("f16x2f32", "block_all_reduce_sum_f16x2_f32"), ("f16x2f16", "block_all_reduce_sum_f16x2_f16"),
                          ("f16x8packf16", "block_all_reduce_sum_f16x8_pack_f16"),
                          ("f16x8packf32", "block_all_reduce_sum_f16x8_pack_f32")], "f16f16_th", None),
    (lambda v: v.bfloat16(), [("bf16bf16", "block_all_reduce_sum_bf16_bf16"), ("bf16f32", "block_all_reduce_sum_bf16_f32"),
                              ("bf16x2f32", "block_all_reduce_sum_bf16x2_f32"),
                              ("bf16x2bf16", "block_all_reduce_sum_bf16x2_bf16"),
                              ("bf16x8packf32", "block_all_reduce_sum_bf16x8_pack_f32"),
                              ("bf16x8packbf16", "block_all_reduce_sum_bf16x8_pack_bf16")], "bf16bf16_th", None),
    (lambda v: v.to(torch.float8_e4m3fn), [("f8e4m3f16", "block_all_reduce_sum_fp8_e4m3_f16"),
                                           ("f8e4m3x16packf16", "block_all_reduce_sum_fp8_e4m3x16_pack_f16")],
     "f8e4m3f16_th", lambda v: v.half()),  # torch.sum has no fp8 (reference :82)
    (lambda v: v.to(torch.float8_e5m2), [("f8e5m2f16", "block_all_reduce_sum_fp8_e5m2_f16"),
                                         ("f8e5m2x16packf16", "block_all_reduce_sum_fp8_e5m2x16_pack_f16")],
     "f8e5m2f16_th", lambda v: v.half()),
    (lambda v: v.to(torch.int8), [("i8i32", "block_all_reduce_sum_i8_i32"),
                                  ("i8x16packi32", "block_all_reduce_sum_i8x16_pack_i32")], "i8i32_th", None),
]


def show(tag, out, ms):
    v = out.item()
    if tag.startswith("i8"):
        print(f"{'out_' + tag:>25}: {v:<15}, time:{ms:.8f}ms")
    else:
        print(f"{'out_' + tag:>25}: {v:<15.8f}, time:{ms:.8f}ms")


def main():
    warmup, iters = (10, 1000) if HAS_GPU else (1, 5)
    sizes = [1024, 2048, 4096]
    json_rows = []
    for S in sizes:
        for K in sizes:
            print("-" * 80)
            print(" " * 40 + f"S={S}, K={K}")
            values = torch.randn((S, K)).to(DEVICE).float()
            for cast, rows, th_tag, th_cast in GROUPS:
                x = cast(values)
                nbytes = x.numel() * x.element_size()
                for tag, name in rows:
                    if lib is None:
                        print(f"{'out_' + tag:>25}: skipped (no GPU: the HIP kernel path has no CPU fallback)")
                        continue
                    fn = getattr(lib, name)
                    out, ms = timed(lambda: fn(x), warmup, iters)
                    show(tag, out, ms)
                    json_rows.append(hbm_row(name, x.shape, ms, nbytes))
                xt = th_cast(x) if th_cast else x
                out, ms = timed(lambda: torch.sum(xt), warmup, iters)
                show(th_tag, out, ms)
                print("-" * 80)
    emit_json(json_rows)


if __name__ == "__main__":
    main()
