"""Where are the wrong elements of probe variant 544 (D = 512 pair kernel, fp32-scaled)? Per launch: rows that differ from the
fp32 reference by > 1e-2, split by 32-row group (= wave pair), by column half (= which wave of the pair wrote it) and by
whether BOTH halves of a row are wrong."""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import host  # noqa: E402

dev = torch.device("cuda:0")
abl = int(sys.argv[1]) if len(sys.argv) > 1 else 544
for (B, H, N) in [(1, 1, 128), (1, 1, 256), (1, 2, 1024)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, 512, dtype=torch.half, device=dev) for _ in range(3))
    r = torch.softmax((q.float() @ k.float().transpose(-1, -2)) / 512 ** 0.5, dim=-1) @ v.float()
    o = torch.zeros_like(q)
    for it in range(6):
        o.zero_()
        host.fa2_variant((8, 0, 0, abl), q, k, v, o)
        torch.cuda.synchronize()
        err = (o.float() - r).abs()
        bad_lo = (err[..., :256].amax(-1) > 1e-2).flatten()
        bad_hi = (err[..., 256:].amax(-1) > 1e-2).flatten()
        rows = torch.nonzero(bad_lo | bad_hi).flatten().tolist()
        grp = Counter((x % N) // 32 % 4 for x in rows)
        both = int((bad_lo & bad_hi).sum())
        print("WHERE abl %d %s launch %d: wrong rows %d (lo half %d, hi half %d, both %d) by 32-row group in the 128-row workgroup %s; first rows %s; max err %.3f"
              % (abl, (B, H, N), it, len(rows), int(bad_lo.sum()), int(bad_hi.sum()), both, dict(grp), rows[:12], err.max().item()), flush=True)
