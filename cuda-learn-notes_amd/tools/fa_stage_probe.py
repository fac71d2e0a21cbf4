"""GPU probe of the `stages` knob of the attention names (stages = 1 runs the load-then-compute kernel -- how
large is the step down to it?): event-timed TFLOPS of stages = 1 and stages = 2 per shape, with the kernel cln_describe names.
  python fa_stage_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu  # noqa: E402

dev = torch.device("cuda:0")
fa = pkg.flash_attn_lib()
for (B, H, N, D) in [(4, 8, 2048, 64), (1, 48, 8192, 64), (4, 8, 2048, 128), (2, 32, 4096, 256), (1, 32, 4096, 512), (1, 16, 4096, 768),
                     (1, 16, 4096, 1024), (1, 16, 4160, 768), (1, 16, 4096, 384), (1, 16, 4096, 640), (4, 8, 1024, 64), (2, 8, 2048, 64)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    fn = fa.flash_attn_mma_stages_split_q_shared_qkv if D <= 256 else fa.flash_attn_mma_stages_split_q_tiling_qkv
    if (B, H, N, D) == (2, 8, 2048, 64):  # the last row: the split-KV rung at its own name
        fn = fa.flash_attn_mma_stages_split_kv
    fl = bu.mha_flops_conventional(B, H, N, D)
    for stages in (1, 2):
        call = lambda: fn(q, k, v, o, stages)
        bu.prewarm(call, 0.2)
        ms = bu.time_region_events(call, 30)
        print("STAGE %-20s stages=%d %8.4f ms %7.1f TF  %s" % ((B, H, N, D), stages, ms, fl / ms * 1e-9, pkg.manifest.describe(fn.__name__, (B, H, N, D), stages)), flush=True)
