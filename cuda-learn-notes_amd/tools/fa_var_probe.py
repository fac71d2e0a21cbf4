"""GPU probe: explicit attention kernel variants of the test-only probe library (host.fa2_variant, `abl` codes of
csrc/flash_attn_probe.hip) against the shipped dispatcher: max-abs-error vs fp32 SDPA on N(0,1) inputs and on a
rescale-regime input (keys amplified, late jumps), then interleaved event timing after a per-variant pre-warm.

  python fa_var_probe.py "542,800,801" "4,8,2048,64;1,48,8192,64" ["543,800" "4,8,2048,128" ...]     (FA_ROUNDS, FA_LAUNCHES)
"""
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
GROUPS = [([int(x) for x in sys.argv[i].split(",")], [tuple(int(x) for x in s.split(",")) for s in sys.argv[i + 1].split(";")])
          for i in range(1, len(sys.argv) - 1, 2)]
ROUNDS = int(os.environ.get("FA_ROUNDS", "3"))
LAUNCHES = int(os.environ.get("FA_LAUNCHES", "30"))


def prewarm(fn, secs=0.25):
    t0 = time.time()
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()


def hard_inputs(B, H, N, D):
    """Keys amplified 4x with a few 6x rows late in the sequence and one spiked (q, k) pair per head: every kernel's
    rescale path (deferred or checked) must fire, also in the LAST tiles."""
    g = torch.Generator(device="cpu").manual_seed(1)
    q = torch.randn(B, H, N, D, generator=g)
    k = torch.randn(B, H, N, D, generator=g) * 4.0
    v = torch.randn(B, H, N, D, generator=g)
    k[:, :, N - 7] *= 1.5
    k[:, :, N // 2 + 3] *= 1.5
    k[:, :, 5] *= 0.1
    q[:, :, 17] = k[:, :, N - 70] * 0.5  # one row whose maximum jumps by a lot near the end
    return (t.to(dev).half() for t in (q, k, v))


def run(ABLS, B, H, N, D):
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    fl = bu.mha_flops_conventional(B, H, N, D)
    name = "flash_attn_mma_stages_split_q_shared_qkv" if D <= 256 else "flash_attn_mma_stages_split_q_tiling_qkv"
    cands = [("shipped", lambda: getattr(fa, name)(q, k, v, o, 2))]
    for abl in ABLS:
        cands.append(("abl %d" % abl, lambda abl=abl: host.fa2_variant((8, 0, 0, abl), q, k, v, o)))
    ok = {}
    for kind in ("randn", "hard"):
        if kind == "hard":
            qq, kk, vv = hard_inputs(B, H, N, D)
            q.copy_(qq), k.copy_(kk), v.copy_(vv)
        hb = min(B * H, 8)  # reference on a bounded number of heads (fp32 SDPA materialises N x N per head)
        ref = F.scaled_dot_product_attention(q.float().flatten(0, 1)[:hb], k.float().flatten(0, 1)[:hb], v.float().flatten(0, 1)[:hb])
        for tag, fn in cands:
            o.zero_()
            try:
                if os.environ.get("FA_TRACE"):
                    print("RUN", (B, H, N, D), kind, tag, flush=True)
                fn()
                torch.cuda.synchronize()
                got = o.float().flatten(0, 1)
                err = (got[:hb] - ref).abs().max().item()
                nan = int(torch.isnan(o).sum().item())
                ok[tag] = ok.get(tag, True) and nan == 0 and err < 1e-2
                print("CHK %s %-6s %-10s max|err| %.3e nan %d %s" % ((B, H, N, D), kind, tag, err, nan, "OK" if nan == 0 and err < 6e-3 else "BAD"), flush=True)
            except Exception as e:
                ok[tag] = False
                print("CHK", (B, H, N, D), kind, tag, "ERR", str(e)[:100], flush=True)
    torch.manual_seed(0)
    for t in (q, k, v):
        t.copy_(torch.randn(B, H, N, D, dtype=torch.half, device=dev))
    cands = [(t, f) for t, f in cands if ok.get(t, False) or os.environ.get("FA_TIME_BAD")]  # FA_TIME_BAD=1: time the ablation variants (wrong by design) too
    for tag, fn in cands:
        prewarm(fn)
    res = {t: [] for t, _ in cands}
    for rnd in range(ROUNDS):
        for tag, fn in cands:
            ms, mn, _ = bu.time_call_events(fn, 5, LAUNCHES)
            res[tag].append(ms)
    base = sum(res["shipped"]) / len(res["shipped"]) if "shipped" in res else None
    for tag, _ in cands:
        ms = sum(res[tag]) / len(res[tag])
        print("FA %s %-10s %8.4f ms %7.1f TF  %+5.1f%%  rounds %s" % ((B, H, N, D), tag, ms, fl / ms * 1e-9, (base / ms - 1) * 100 if base else 0.0,
                                                                  " ".join("%.4f" % r for r in res[tag])), flush=True)


for ABLS, SHAPES in GROUPS:
    for shp in SHAPES:
        run(ABLS, *shp)
