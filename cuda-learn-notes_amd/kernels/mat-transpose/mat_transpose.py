"""mat-transpose bench -- same rows/tags as reference kernels/mat-transpose/mat_transpose.py:69-95, including its
`out.T.equal(x)` column. No GPU: only the torch.transpose_copy row runs, on CPU."""
import os
import sys
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, emit_json, hbm_row, package, timed  # noqa: E402

lib = package().load("mat_transpose") if HAS_GPU else None
ROWS = [("f32_col2row", "mat_transpose_f32_col2row"), ("f32_row2col", "mat_transpose_f32_row2col"),
        ("f32_col2row(2d)", "mat_transpose_f32_col2row2d"), ("f32_row2col(2d)", "mat_transpose_f32_row2col2d"),
        ("f32_diagnonal", "mat_transpose_f32_diagonal2d"), ("f32x4_col2row", "mat_transpose_f32x4_col2row"),
        ("f32x4_row2col", "mat_transpose_f32x4_row2col"), ("f32x4_col2row(2d)", "mat_transpose_f32x4_col2row2d"),
        ("f32x4_row2col(2d)", "mat_transpose_f32x4_row2col2d"),
        ("f32x4_shared_col2row(2d)", "mat_transpose_f32x4_shared_col2row2d"),
        ("f32x4_shared_row2col(2d)", "mat_transpose_f32x4_shared_row2col2d"),
        ("f32x4_shared_bcf_col2row(2d)", "mat_transpose_f32x4_shared_bcf_col2row2d"),
        ("f32x4_shared_bcf_row2col(2d)", "mat_transpose_f32x4_shared_bcf_row2col2d")]


def main():
    warmup, iters = (10, 1000) if HAS_GPU else (1, 3)
    json_rows = []
    for M in (1024, 2048, 4096):
        for N in (1024, 2048, 4096):
            print("-" * 130)
            print(" " * 55 + f"M={M}, N={N}")
            x = torch.randn((M, N)).to(DEVICE).float().contiguous()
            y = torch.randn((N, M)).to(DEVICE).float().contiguous()
            nb = 2 * x.numel() * 4
            for tag, name in ROWS:
                if lib is None:
                    print(f"{'out_' + tag:>35}: skipped (no GPU: the HIP kernel path has no CPU fallback)")
                    continue
                y.fill_(0)
                fn = getattr(lib, name)
                _, ms = timed(lambda: fn(x, y), warmup, iters)
                vals = [round(v, 8) for v in y.flatten()[:3].tolist()]
                print(f"{'out_' + tag:>35}: {vals}, validate {str(y.T.equal(x)):<5}, time:{ms:.8f}ms")
                json_rows.append(hbm_row(name, x.shape, ms, nb))
            _, ms = timed(partial(torch.transpose_copy, x, dim0=0, dim1=1, out=y), warmup, iters)
            print(f"{'out_f32_th':>35}: {[round(v, 8) for v in y.flatten()[:3].tolist()]}, validate {str(y.T.equal(x)):<5}, time:{ms:.8f}ms")
    emit_json(json_rows)


if __name__ == "__main__":
    main()
