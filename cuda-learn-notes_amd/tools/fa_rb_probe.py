"""GPU probe for the register-blocked FlashAttention kernel (csrc/flash_attn_rb.cuh): interleaved A/B of its option
sets against the shipped dispatcher (ping-pong kernel) and torch SDPA, with a max-abs-error check against fp32 SDPA.

  python fa_rb_probe.py [quick]          -> stdout (tee into gpurun_out/, the summary goes to profiles/)

Timing: every candidate is pre-warmed for 0.3 s, then 3 interleaved rounds of 20 event-timed launches each (cdna guide
rules 24/9: within-probe interleaved rounds; report mean and best)."""
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
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"

NAMES = {400: "builtin", 401: "asmqk", 402: "builtin no-pre", 403: "builtin unpinned", 404: "builtin bc32", 405: "asmqk bc32",
         406: "asmqk pd2", 407: "asmqk no-defer", 408: "asmqk no-pre", 409: "builtin no-defer", 410: "asmqk ABL no-dma",
         411: "asmqk ABL no-exp", 412: "asmqk ABL no-dma no-exp", 420: "builtin half-step", 421: "asmqk half-step",
         422: "asmqk half-step pd2"}
NAMES128 = {400: "builtin", 401: "asmqk", 405: "builtin unpinned", 407: "asmqk no-defer", 408: "asmqk pd2", 409: "builtin no-defer",
            410: "asmqk ABL no-dma", 411: "asmqk ABL no-exp", 420: "builtin half-step", 421: "asmqk half-step"}
VARIANTS = {64: [], 128: []} if os.environ.get("RB_FEW") else {64: [400, 401, 402, 403, 404, 405, 406, 408, 410, 411, 412], 128: [400, 401, 405, 408, 410, 411]}
PP = {64: [(500, "pre stagger bc128"), (508, "pre stagger pd8"), (509, "pre stagger pd16"), (510, "pre pd16"), (511, "pre stagger bc64 pd8")],
      128: [(500, "pre kpre bc64"), (508, "pre pd8"), (509, "pre pd16")]}
SHAPES = [(4, 8, 2048, 64), (1, 48, 8192, 64), (4, 8, 2048, 128), (2, 32, 4096, 128)]
if quick:
    SHAPES = [(4, 8, 2048, 64), (4, 8, 2048, 128)]


def prewarm(fn, secs=0.3):
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
    names = NAMES if D == 64 else NAMES128
    cands = [("shipped (dispatcher)", lambda: fa.flash_attn_mma_stages_split_q_shared_qkv(q, k, v, o, 2)),
             ("sdpa", lambda: F.scaled_dot_product_attention(q, k, v))]
    for abl in VARIANTS[D]:
        cands.append(("rb %d %s" % (abl, names[abl]), lambda abl=abl: host.fa2_variant((4, 0, 0, abl), q, k, v, o)))
    for abl, nm in PP[D]:
        cands.append(("pp %d %s" % (abl, nm), lambda abl=abl: host.fa2_variant((8, 0, 0, abl), q, k, v, o)))
    ok = {}
    for tag, fn in cands:
        if tag == "sdpa":
            continue
        o.zero_()
        try:
            fn()
            torch.cuda.synchronize()
            err = (o.float() - ref).abs().max().item()
            ok[tag] = True
            good = err < 3e-3 or "ABL" in tag
            print("CHK %s %-28s max|err| %.3e %s" % ((B, H, N, D), tag, err, "OK" if good else "BAD"), flush=True)
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
        mn = min(r[1] for r in res[tag])
        print("FA %s %-28s %8.4f ms %7.1f TF (best launch %7.1f TF)  rounds %s" % (
            (B, H, N, D), tag, ms, fl / ms * 1e-9, fl / mn * 1e-9, " ".join("%.4f" % r[0] for r in res[tag])), flush=True)
