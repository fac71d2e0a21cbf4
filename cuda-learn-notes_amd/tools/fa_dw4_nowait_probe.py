"""Round-5 probe (timing only, the ablated forms write garbage): the one-wave-per-SIMD attention kernel with its K / V requests issued but never waited for
(codes + 1024) and not issued at all (+ 4), against the production options -- is the 20 % the requests cost ISSUE time or LANDING time?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host  # noqa: E402

dev = torch.device("cuda:0")
CODES = {640: (1444, 2468, 1448), 768: (1540, 2564, 1544), 1024: (1540, 2564, 1544)}
for (B, H, N, D) in [(1, 16, 4096, 640), (1, 16, 4096, 768), (1, 16, 4096, 1024)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    fl = bu.mha_flops_conventional(B, H, N, D)
    for rnd in range(2):
        for c in CODES[D]:
            fn = (lambda a: lambda: host.fa2_variant((4, 0, 0, a), q, k, v, o))(c)
            bu.prewarm(fn, 0.15)
            ms = bu.time_region_events(fn, 20)
            print("NOWAIT %s r%d dw4 %d %8.4f ms %7.1f TF" % ((B, H, N, D), rnd, c, ms, fl / ms * 1e-9), flush=True)
