"""GPU probe: AMD's ck_tile FMHA forward kernels (csrc/fa2_vendor_ck.hip -- the kernel family FlashAttention-2-ROCm / aiter
dispatch to) beside ours and torch SDPA on the attention bench shapes: max-abs-error against an fp32 reference, then
event-timed TFLOPS (4 B H N^2 D) after a pre-warm, ONE region of back-to-back launches each.
  python fa_ck_probe.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu  # noqa: E402

dev = torch.device("cuda:0")
fa = pkg.flash_attn_lib()
ck = pkg.load("fa2_vendor_ck").cln_fa2_ck_tile_fwd
for (B, H, N, D) in [(4, 8, 2048, 64), (2, 24, 4096, 64), (1, 48, 8192, 64), (4, 8, 2048, 128), (2, 32, 4096, 128), (1, 24, 8192, 128)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    ref = F.scaled_dot_product_attention(q[:, :2].float(), k[:, :2].float(), v[:, :2].float())
    fl = bu.mha_flops_conventional(B, H, N, D)
    rows = [("ours", lambda o: fa.flash_attn_mma_stages_split_q_shared_qkv(q, k, v, o, 2)),
            ("ck_tile async", lambda o: ck(q, k, v, o, 0))]
    if D == 128:
        rows.append(("ck_tile v3 (gfx950)", lambda o: ck(q, k, v, o, 3)))
    rows.append(("torch SDPA", None))
    for tag, fn in rows:
        o = torch.zeros_like(q)
        call = (lambda: fn(o)) if fn else (lambda: F.scaled_dot_product_attention(q, k, v))
        out = call()
        torch.cuda.synchronize()
        got = o if fn else out
        err = (got[:, :2].float() - ref).abs().max().item()
        bu.prewarm(call, 0.25)
        iters = 200 if N <= 2048 else 50
        ms = bu.time_region_events(call, iters)
        print("CK %-20s %-20s max|err| %.3e  %8.4f ms  %7.1f TF" % ((B, H, N, D), tag, err, ms, fl / ms * 1e-9), flush=True)
