"""GPU probe of the single-stage (`stages = 1`) forms of the attention kernels (round 4): where in the iteration the
one-burst request + wait of the next tile goes (M16X_ONE_POS 0..3), against the stage-2 kernel and against the 4-wave
load-then-compute kernel that `stages = 1` ran until round 3. Every form is checked bit-identical to stage 2.
  python fa_one_stage_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host  # noqa: E402

dev = torch.device("cuda:0")
fa = pkg.flash_attn_lib()


def run(tag, shape, call, o, ref):
    bu.prewarm(call, 0.2)
    ms = bu.time_region_events(call, 100 if shape[2] <= 2048 else 25)
    torch.cuda.synchronize()
    fl = bu.mha_flops_conventional(*shape)
    same = "-" if ref is None else str(bool(torch.equal(o, ref)))
    print("ONE %-20s %-44s %8.4f ms %7.1f TF  bit-identical to stage 2: %s" % (shape, tag, ms, fl / ms * 1e-9, same), flush=True)


for shape, codes in (((4, 8, 2048, 64), (170, 171, 172, 173)), ((4, 8, 2048, 128), (170, 171, 172, 173)), ((1, 48, 8192, 64), (180, 181, 182, 183))):
    B, H, N, D = shape
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    ref, o = torch.zeros_like(q), torch.zeros_like(q)
    fn = fa.flash_attn_mma_stages_split_q_shared_qkv
    run("stages=2 (product)", shape, lambda: fn(q, k, v, ref, 2), ref, None)
    run("stages=1 (product)", shape, lambda: fn(q, k, v, o, 1), o, ref)
    for c in codes:
        o.zero_()
        run("m16x one-stage, burst position %d" % (c % 10), shape, lambda: host.fa2_variant((8, 0, 0, 800 + c), q, k, v, o), o, ref)
    o.zero_()
    run("4-wave load-then-compute (rounds 1-3)", shape, lambda: host.fa2_variant((4, 0, 0, 90), q, k, v, o), o, None)
for shape in ((2, 32, 4096, 256), (1, 32, 4096, 512)):
    B, H, N, D = shape
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    ref, o = torch.zeros_like(q), torch.zeros_like(q)
    fn = fa.flash_attn_mma_stages_split_q_shared_qkv if D <= 256 else fa.flash_attn_mma_stages_split_q_tiling_qkv
    run("stages=2 (product)", shape, lambda: fn(q, k, v, ref, 2), ref, None)
    run("stages=1 (product: one burst per tile)", shape, lambda: fn(q, k, v, o, 1), o, ref)
    o.zero_()
    run("pair kernel: burst at the top of phase B", shape, lambda: host.fa2_variant((8, 0, 0, 547), q, k, v, o), o, ref)
    if D == 512:
        o.zero_()
        run("round 3: wait after every piece", shape, lambda: host.fa2_variant((8, 0, 0, 546), q, k, v, o), o, ref)
    if D == 256:
        o.zero_()
        run("4-wave load-then-compute (rounds 1-3)", shape, lambda: host.fa2_variant((4, 0, 0, 90), q, k, v, o), o, None)
