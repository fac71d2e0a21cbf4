"""GPU probe for the one-wave-per-SIMD attention kernel (csrc/flash_attn_w4.cuh): schedule variants (abl 600+) and
ablations (610+) against the shipped dispatcher and torch SDPA, max-abs-error vs fp32 SDPA.
  python fa_w4_probe.py [abl,abl,...] ["B,H,N,D;..."]"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host  # noqa: E402

dev = torch.device("cuda:0")
fa = pkg.flash_attn_lib()
ABLS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "600,601,602").split(",")]
SHAPES = [tuple(int(x) for x in s.split(",")) for s in (sys.argv[2] if len(sys.argv) > 2 else "4,8,2048,64;1,48,8192,64;4,8,2048,128;2,32,4096,128").split(";")]


def prewarm(fn, secs=0.2):
    t0 = time.time()
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()


for (B, H, N, D) in SHAPES:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float())
    fl = bu.mha_flops_conventional(B, H, N, D)
    cands = [("shipped (dispatcher)", lambda: fa.flash_attn_mma_stages_split_q_shared_qkv(q, k, v, o, 2)),
             ("sdpa", lambda: F.scaled_dot_product_attention(q, k, v))]
    for abl in ABLS:
        cands.append(("w4 %d" % abl, lambda abl=abl: host.fa2_variant((4, 0, 0, abl), q, k, v, o)))
    for abl in [int(x) for x in os.environ.get("FA_PP2", "").split(",") if x]:
        if (abl < 600 and (abl not in (545, 546) or (D == 64 and N % 512 == 0))) or (abl >= 600 and D == 64 and N % (256 if abl >= 710 else 512) == 0):  # 5xx: variants of the shipped ping-pong kernel
            cands.append(("pp64rows %d" % abl, lambda abl=abl: host.fa2_variant((8, 0, 0, abl), q, k, v, o)))
    ok = {}
    for tag, fn in cands:
        if tag == "sdpa":
            continue
        o.zero_()
        try:
            fn()
            torch.cuda.synchronize()
            err = (o.float() - ref).abs().max().item()
            nan = int(torch.isnan(o).sum().item())
            ok[tag] = True
            abl_n = int(tag.split()[1]) if tag.startswith("w4") else 0
            print("CHK %s %-22s max|err| %.3e nan %d %s" % ((B, H, N, D), tag, err, nan, "OK" if (err < 6e-3 and nan == 0) or abl_n >= 610 else "BAD"), flush=True)
        except Exception as e:
            ok[tag] = False
            print("CHK", tag, "ERR", str(e)[:100], flush=True)
    cands = [(t, f) for t, f in cands if ok.get(t, True)]
    for tag, fn in cands:
        prewarm(fn, 0.2)
    res = {t: [] for t, _ in cands}
    for rnd in range(3):
        for tag, fn in cands:
            ms, mn, _ = bu.time_call_events(fn, 3, 20)
            res[tag].append((ms, mn))
    for tag, _ in cands:
        ms = sum(r[0] for r in res[tag]) / len(res[tag])
        print("FA %s %-22s %8.4f ms %7.1f TF  rounds %s" % ((B, H, N, D), tag, ms, fl / ms * 1e-9, " ".join("%.4f" % r[0] for r in res[tag])), flush=True)
