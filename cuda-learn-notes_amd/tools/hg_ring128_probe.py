"""GPU probe: the 128x128 (4-wave) and 128x256 / 256x128 (8-wave) multi-stage rings at every (BK, stages) that is instantiated --
which BK should ring_pick choose per stage count? (128x128 at stages 3 slower than at stages 2.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host  # noqa: E402

dev = torch.device("cuda:0")
TILES = {0: "128x128 4w", 3: "128x256 8w", 2: "256x128 8w", 1: "256x256 8w", 6: "64x128 4w", 7: "64x64 4w"}
for M in (4096, 2048):
    N = K = M
    torch.manual_seed(1)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    stride = bu.make_block_swizzle_stride(N, K)
    fl = bu.hgemm_flops(M, N, K)
    for tile in (0, 3, 2, 1, 6, 7):
        for bk in (64, 32):
            for S in (2, 3, 4, 5):
                call = lambda: host.hgemm_variant(0, 0, tile, bk, S, a, b, c, 1, stride)
                try:
                    bu.prewarm(call, 0.15)
                except RuntimeError:
                    continue
                ms = bu.time_region_events(call, 60)
                print("RING %5d^3 %-11s BK=%d stages=%d LDS %3d KiB %8.4f ms %7.1f TF" %
                      (M, TILES[tile], bk, S, S * (int(TILES[tile].split('x')[0]) + int(TILES[tile].split('x')[1].split()[0])) * bk * 2 // 1024, ms, fl / ms * 1e-9), flush=True)
