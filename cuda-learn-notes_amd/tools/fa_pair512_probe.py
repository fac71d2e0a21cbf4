"""GPU probe for the D = 512 pair kernel on 16x16x32 MFMAs (csrc/flash_attn_m16.cuh fa2_fwd_m16_pair_kernel, probe variants
540 = Q pre-scaled in fp16, 544 = scores scaled in fp32): DESIGN r2 section 9.1 reported variant 544 as WRONG ("sparse large
errors, cause not found"). Here: max-abs-error against a chunked fp32 reference on EVERY head of several shapes, repeated on
fresh inputs, plus bit-repeatability of repeated launches -- a layout / hazard bug shows on every run, a box fault does not."""
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


def ref32(q, k, v):
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    sc = 1.0 / (q.shape[-1] ** 0.5)
    for b in range(q.shape[0]):
        for h in range(q.shape[1]):
            out[b, h] = torch.softmax((q[b, h].float() @ k[b, h].float().t()) * sc, dim=-1) @ v[b, h].float()
    return out


for (B, H, N) in [(1, 32, 4096), (2, 3, 256), (1, 8, 1024), (3, 5, 640), (1, 64, 2048)]:
    for seed in (0, 1, 2):
        torch.manual_seed(seed)
        q, k, v = (torch.randn(B, H, N, 512, dtype=torch.half, device=dev) for _ in range(3))
        if seed == 2:  # amplified keys: the rescale path fires
            k = (k.float() * torch.linspace(0.5, 3.0, N, device=dev).view(1, 1, N, 1)).half()
        r = ref32(q, k, v)
        o = torch.zeros_like(q)
        rows = []
        for tag, fn in (("shipped", lambda: fa.flash_attn_mma_stages_split_q_tiling_qkv(q, k, v, o, 2)),
                        ("540 pre-scaled", lambda: host.fa2_variant((8, 0, 0, 540), q, k, v, o)),
                        ("544 fp32-scaled", lambda: host.fa2_variant((8, 0, 0, 544), q, k, v, o))):
            o.zero_()
            fn()
            torch.cuda.synchronize()
            first = o.clone()
            err = (first.float() - r).abs().amax(dim=(2, 3)).flatten()
            bad = 0
            for _ in range(20):
                o.zero_()
                fn()
                bad += 0 if torch.equal(o, first) else 1
            rows.append("%s max|err| %.3e (worst head %d) mismatching relaunches %d/20" % (tag, err.max().item(), int(err.argmax()), bad))
        print("P512 %s seed %d | %s" % ((B, H, N), seed, " | ".join(rows)), flush=True)
B, H, N = 1, 32, 4096
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, N, 512, dtype=torch.half, device=dev) for _ in range(3))
o = torch.zeros_like(q)
fl = bu.mha_flops_conventional(B, H, N, 512)
cands = [("shipped", lambda: fa.flash_attn_mma_stages_split_q_tiling_qkv(q, k, v, o, 2)), ("540", lambda: host.fa2_variant((8, 0, 0, 540), q, k, v, o)),
         ("544", lambda: host.fa2_variant((8, 0, 0, 544), q, k, v, o))]
for tag, fn in cands:
    bu.prewarm(fn, 0.25)
for rnd in range(3):
    for tag, fn in cands:
        ms = bu.time_region_events(fn, 20)
        print("P512 time r%d %-8s %8.4f ms %7.1f TF" % (rnd, tag, ms, fl / ms * 1e-9), flush=True)
