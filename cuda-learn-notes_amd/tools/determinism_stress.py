"""GPU stress: every dispatched attention / HGEMM kernel is deterministic, so repeated launches on the same inputs must be
BIT-identical. Launch each shape REPS times, compare every output with the first (and the first with a chunked fp32
reference): a mismatch is a race or a hardware fault, never rounding.   python determinism_stress.py [reps]"""
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
hg = pkg.hgemm_lib()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SHAPES = [(2, 96, 256, 256), (4, 8, 2048, 256), (4, 8, 2048, 64), (4, 8, 2048, 128), (1, 48, 8192, 64), (1, 32, 4096, 512),
          (1, 16, 2048, 768), (1, 16, 2048, 1024), (1, 16, 2048, 384), (2, 8, 2048, 64), (1, 2, 256, 96)]
if len(sys.argv) > 2:
    SHAPES = [tuple(int(x) for x in s.split(",")) for s in sys.argv[2].split(";")]
bad_total = 0
for (B, H, N, D) in SHAPES:
    torch.manual_seed(7)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    fn = fa.flash_attn_mma_stages_split_q_shared_qkv if D <= 256 else fa.flash_attn_mma_stages_split_q_tiling_qkv
    fn(q, k, v, o, 2)
    torch.cuda.synchronize()
    first = o.clone()
    hb = min(B * H, 4)
    ref = F.scaled_dot_product_attention(q.float().flatten(0, 1)[:hb], k.float().flatten(0, 1)[:hb], v.float().flatten(0, 1)[:hb])
    err = (first.float().flatten(0, 1)[:hb] - ref).abs().max().item()
    bad, worst = 0, 0.0
    for r in range(REPS):
        o.zero_()
        fn(q, k, v, o, 2)
        if not torch.equal(o, first):
            bad += 1
            worst = max(worst, (o.float() - first.float()).abs().max().item())
    bad_total += bad
    print("DET fa %s err-vs-fp32 %.2e  mismatching launches %d / %d  worst |diff| %.3e  %s" % ((B, H, N, D), err, bad, REPS, worst,
                                                                                              pkg.manifest.describe(fn.__name__, (B, H, N, D), 2)[:40]), flush=True)
# squares, then (round 5) shapes that take the one-launch split-K (2 splits, fix-up by the last-arriving workgroup: the sum order must not depend on
# which workgroup arrives last), the two-launch split-K and the tail split
for S in (4096, 2048, 3072, 2560, (512, 8192, 8192), (640, 5120, 5120), (1024, 1024, 16384), (4864, 4864, 4864)):
    M_, N_, K_ = S if isinstance(S, tuple) else (S, S, S)
    torch.manual_seed(3)
    a = torch.randn(M_, K_, dtype=torch.half, device=dev)
    b = torch.randn(K_, N_, dtype=torch.half, device=dev)
    c = torch.zeros(M_, N_, dtype=torch.half, device=dev)
    st = bu.make_block_swizzle_stride(N_, K_)
    for name in ("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem",):
        fn = getattr(hg, name)
        fn(a, b, c, 2, True, st)
        torch.cuda.synchronize()
        first = c.clone()
        bad = 0
        for r in range(REPS):
            c.zero_()
            fn(a, b, c, 2, True, st)
            bad += 0 if torch.equal(c, first) else 1
        bad_total += bad
        print("DET hgemm %s %s mismatching launches %d / %d  %s" % ("x".join(map(str, (M_, N_, K_))), name[-28:], bad, REPS,
                                                                     pkg.manifest.describe(name, (M_, N_, K_), 2)[-70:]), flush=True)
# round 6: the LDS-DMA f32-MFMA sgemm in its three tile forms (64x128, 128x128, 256x128): ring slots re-used three stages later, the barrier in
# the middle of a stage -- a request that overtook a reader would show here
sg = pkg.load("sgemm")
for (M_, N_, K_) in ((3072, 3072, 2048), (4096, 4096, 4096), (8192, 8192, 1024), (1024, 1024, 8192)):
    torch.manual_seed(5)
    a = torch.randn(M_, K_, device=dev)
    b = torch.randn(K_, N_, device=dev)
    c = torch.zeros(M_, N_, device=dev)
    fn = sg.sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages
    fn(a, b, c, 2, True, 256)
    torch.cuda.synchronize()
    first = c.clone()
    sel = torch.tensor([0, M_ // 2 + 1, M_ - 1], device=dev)
    truth = a[sel].double() @ b.double()
    err = ((first[sel].double() - truth).abs().max() / truth.abs().max()).item()
    bad = 0
    for r in range(min(REPS, 100)):
        c.fill_(float("nan"))
        fn(a, b, c, 3 if r & 1 else 2, bool(r & 2), 256)
        bad += 0 if torch.equal(c, first) else 1
    bad_total += bad
    print("DET sgemm %s sampled rel err vs fp64 %.2e  mismatching launches %d / %d (stages / swizzle knobs alternating)" % ("x".join(map(str, (M_, N_, K_))), err, bad, min(REPS, 100)), flush=True)
print("DET total mismatches", bad_total)
