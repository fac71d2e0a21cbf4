"""histogram demo -- same input and printed rows as reference kernels/histogram/histogram.py:22-33.
No GPU: prints the torch.bincount column only (the HIP kernel path has no CPU fallback)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _common import DEVICE, HAS_GPU, package  # noqa: E402

lib = package().load("histogram") if HAS_GPU else None


def main():
    a = torch.tensor(list(range(10)) * 1000, dtype=torch.int32).to(DEVICE)
    for tag, name in (("h_i32  ", "histogram_i32"), ("h_i32x4", "histogram_i32x4")):
        print("-" * 80)
        if lib is None:
            print(f"{tag}: skipped (no GPU: the HIP kernel path has no CPU fallback)")
            continue
        h = getattr(lib, name)(a)
        for i in range(h.shape[0]):
            print(f"{tag} {i}: {h[i]}")
    print("-" * 80)
    h = torch.bincount(a.long())
    for i in range(h.shape[0]):
        print(f"h_th    {i}: {h[i]}")
    print("-" * 80)


if __name__ == "__main__":
    main()
