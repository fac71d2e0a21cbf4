"""A/B of the big-D kernel's fragment-prefetch forms at config C5 ([1,32,4096,512]) and [1,8,8192,512]."""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host
dev = torch.device("cuda:0")
fa = pkg.flash_attn_lib()
VARIANTS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "201,203,204,205,206").split(",")]
SHAPES = [tuple(int(x) for x in s.split(",")) for s in sys.argv[2].split(";")] if len(sys.argv) > 2 else [(1, 32, 4096, 512), (1, 8, 8192, 512)]
for (B, H, N, D) in SHAPES:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float())
    fl = bu.mha_flops_conventional(B, H, N, D)
    prod = fa.flash_attn_mma_stages_split_q_tiling_qkv if D > 256 else fa.flash_attn_mma_stages_split_q_shared_qkv
    cands = [("production", lambda: prod(q, k, v, o, 2)), ("sdpa", lambda: F.scaled_dot_product_attention(q, k, v))]
    for a in VARIANTS:
        cands.append(("abl %d" % a, (lambda a: lambda: host.fa2_variant((4, 0, 15, a), q, k, v, o))(a)))
    for tag, fn in cands:
        if tag == 'sdpa':
            continue
        o.zero_()
        try:
            fn(); torch.cuda.synchronize()
            print("CHK %s %-12s max|err| %.3e" % ((B, H, N, D), tag, (o.float() - ref).abs().max().item()), flush=True)
        except Exception as e:
            print("CHK", tag, "ERR", str(e)[:100], flush=True)
    for rnd in range(2):
        for tag, fn in cands:
            try:
                ms, mn, _ = bu.time_call_events(fn, 2, 8)
                print("FA %s r%d %-12s %8.4f ms %7.1f TF" % ((B, H, N, D), rnd, tag, ms, fl / ms * 1e-9), flush=True)
            except Exception as e:
                print("FA", tag, "ERR", str(e)[:100], flush=True)
