"""GPU probe for the v2 FlashAttention kernel: interleaved A/B of variants vs v1 and torch SDPA, with a
max-abs-error check against fp32 SDPA. python fa_probe.py [quick]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host  # noqa: E402

dev = torch.device("cuda:0")
fa = pkg.flash_attn_lib()
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"

VARIANTS = {  # (waves, v_transposed, OPT mask, ablation | 100 = v3 | 300 = v4)
    64: [(8, 0, 13, 0), (8, 0, 13, 100), (8, 0, 16397, 100), (4, 0, 16397, 100)],
    128: [(8, 0, 15, 0)],
}
SHAPES = [(4, 8, 2048, 64), (1, 48, 8192, 64)]
if quick:
    SHAPES = SHAPES[:2]

for (B, H, N, D) in SHAPES:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float())
    fl = bu.mha_flops_conventional(B, H, N, D)
    cands = [("v1 s1", lambda: fa.flash_attn_mma_stages_split_q_shared_qkv(q, k, v, o, 1)),
             ("v1 s2", lambda: fa.flash_attn_mma_stages_split_q_shared_qkv(q, k, v, o, 2)),
             ("sdpa", lambda: F.scaled_dot_product_attention(q, k, v))]
    for var in VARIANTS[D]:
        cands.append(("v2 nw%d opt%d abl%d" % (var[0], var[2], var[3]),
                      lambda var=var: host.fa2_variant(var, q, k, v, o)))
    # correctness of the non-ablated variants
    for tag, fn in cands:
        if tag == "sdpa" or "abl" in tag and not (tag.endswith("abl0") or tag.endswith("abl100") or tag.endswith("abl300")):
            continue
        o.zero_()
        try:
            fn()
            torch.cuda.synchronize()
            err = (o.float() - ref).abs().max().item()
            print("CHK %s %-22s max|err| %.3e %s" % ((B, H, N, D), tag, err, "OK" if err < 2e-3 else "BAD"), flush=True)
        except Exception as e:
            print("CHK", tag, "ERR", str(e)[:100], flush=True)
    for rnd in range(2):
        for tag, fn in cands:
            try:
                ms, mn, _ = bu.time_call_events(fn, 3, 15)
                print("FA %s r%d %-22s %8.4f ms %7.1f TF (best %7.1f)" % ((B, H, N, D), rnd, tag, ms, fl / ms * 1e-9,
                                                                       fl / mn * 1e-9), flush=True)
            except Exception as e:
                print("FA", tag, "ERR", str(e)[:100], flush=True)
