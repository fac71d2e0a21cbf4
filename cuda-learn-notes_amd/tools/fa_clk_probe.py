"""Sample rocm-smi clocks / power while an attention kernel loops (DVFS view). python fa_clk_probe.py"""
import os, subprocess, sys, threading, time, re
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host
dev = torch.device("cuda:0")
fa = pkg.flash_attn_lib()


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        m = re.search(r'sclk clock speed:": "\((\d+)Mhz\)', out); p = re.search(r'Power \(W\)": "([\d.]+)', out)
        return "sclk %s MHz  %s W" % (m.group(1) if m else "?", p.group(1) if p else "?")
    except Exception as e:
        return "smi error %s" % e


ABLATE = os.environ.get("FA_CLK_ABLATE")  # energy ablations of the shipped C4 kernel (probe variants 520..527)
R3 = os.environ.get("FA_CLK_R3")  # round 3: the sum-checked kernel and its opaque-operand ablations (probe variants 853, 870..884)
for (B, H, N, D) in (((4, 8, 2048, 64), (2, 24, 4096, 64)) if R3 else ((4, 8, 2048, 64),) if ABLATE else ((4, 8, 2048, 64), (4, 8, 2048, 128), (2, 32, 4096, 128))):
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    z = torch.zeros_like(q)
    o = torch.zeros_like(q)
    fl = bu.mha_flops_conventional(B, H, N, D)
    cands = [("shipped", lambda: fa.flash_attn_mma_stages_split_q_shared_qkv(q, k, v, o, 2)),
             ("shipped zeros", lambda: fa.flash_attn_mma_stages_split_q_shared_qkv(z, z, z, o, 2)),
             ("w4 %d" % (608 if D == 64 else 600), lambda: host.fa2_variant((4, 0, 0, 608 if D == 64 else 600), q, k, v, o)),
             ("w4 skeleton 615", lambda: host.fa2_variant((4, 0, 0, 615), q, k, v, o)),
             ("w4 skeleton zeros", lambda: host.fa2_variant((4, 0, 0, 615), z, z, z, o))]
    if R3:
        names = {853: "shipped kernel (probe 853)", 870: "no K fragment reads", 871: "no V fragment reads", 872: "no K, V reads", 873: "no exponentials", 874: "no LDS-DMA",
                 875: "no K, V reads, no DMA", 876: "skeleton (no reads, exp, DMA)", 880: "skeleton, no barriers", 882: "snake order", 884: "M V M V order"}
        cands = [("shipped", cands[0][1]), ("shipped zeros", cands[1][1])] + [(names[a], lambda a=a: host.fa2_variant((8, 0, 0, a), q, k, v, o)) for a in names]
    if ABLATE:
        names = {500: "shipped (variant 500)", 520: "no DMA", 521: "no exp", 522: "no fragment reads", 523: "no PV MFMA", 524: "no QK MFMA",
                 525: "no MFMA", 526: "no exp, no fragment reads", 527: "no exp, no reads, no DMA"}
        cands = [(names[a], lambda a=a: host.fa2_variant((8, 0, 0, a), q, k, v, o)) for a in (500, 520, 521, 522, 523, 524, 525, 526, 527)]
    for tag, fn in cands:
        res = {}

        def sampler():
            time.sleep(1.0)
            res["smi"] = smi()

        th = threading.Thread(target=sampler)
        th.start()
        t0 = time.time(); n = 0
        while time.time() - t0 < 2.2:
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
            n += 200
        dt = time.time() - t0
        th.join()
        print("%s %-28s %7.1f TF sustained | %s" % ((B, H, N, D), tag, fl * n / dt * 1e-12, res.get("smi")), flush=True)
