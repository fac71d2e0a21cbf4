"""Round-5 probe: do the four waves' LDS-DMA requests of the one-wave-per-SIMD attention kernel (csrc/flash_attn_dw4.cuh) collide at the CU's address unit?
After every loop barrier wave w idles 16 w (DW4_SKEW4) or 32 w (DW4_SKEW8) clocks, so the requests arrive one piece-time apart instead of together.
Bit-identity to the production form first, then interleaved timing rounds."""
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
prod = fa.flash_attn_mma_stages_split_q_tiling_qkv
CODES = {640: (1444, 1700, 1956, 2133, 2144, 2132), 768: (1540, 1796, 2052, 2133, 2144, 2132, 2123), 1024: (1540, 1796, 2052)}  # production options, + 256 (16 w clocks), + 512 (32 w clocks); 2100 + 10 KPF + VPF = K / V fragments in flight
for D in (640, 768, 1024):
    torch.manual_seed(0)
    q, k, v = (torch.randn(1, 3, 1152, D, dtype=torch.half, device=dev) for _ in range(3))
    outs = []
    for c in CODES[D]:
        o = torch.zeros_like(q)
        host.fa2_variant((4, 0, 0, c), q, k, v, o)
        outs.append(o)
    torch.cuda.synchronize()
    print("SKEW D=%d bit-identical to the production form: %s" % (D, [torch.equal(x, outs[0]) for x in outs[1:]]), flush=True)
for (B, H, N, D) in [(1, 16, 4096, 640), (1, 16, 4096, 768), (1, 16, 4096, 1024), (1, 8, 8192, 1024)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    fl = bu.mha_flops_conventional(B, H, N, D)
    cands = [("production name", lambda: prod(q, k, v, o, 2))] + [("dw4 %d" % c, (lambda a: lambda: host.fa2_variant((4, 0, 0, a), q, k, v, o))(c)) for c in CODES[D]]
    for rnd in range(3):
        for tag, fn in cands:
            bu.prewarm(fn, 0.15)
            ms = bu.time_region_events(fn, 20)
            print("SKEW %s r%d %-18s %8.4f ms %7.1f TF" % ((B, H, N, D), rnd, tag, ms, fl / ms * 1e-9), flush=True)
