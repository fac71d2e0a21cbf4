"""GPU probe: the one-wave-per-SIMD HGEMM kernel (no-store form, schedule 4) with parts removed -- fragment reads, LDS-DMA, both, barriers -- and with
every LDS-DMA piece re-reading the SAME KiB (probe 118: the instruction count and the LDS writes of the real kernel without L2 / fabric traffic), 4096^3,
interleaved rounds of event-timed launches. Results of the ablated forms are garbage by design.   python hg_dma_ablation_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host
dev = torch.device("cuda:0")
S = 4096
a = torch.randn(S, S, dtype=torch.half, device=dev); b = torch.randn(S, S, dtype=torch.half, device=dev); c = torch.zeros(S, S, dtype=torch.half, device=dev)
cands = [("no-store base (104)", 104), ("no reads (111)", 111), ("no DMA (112)", 112), ("no reads no DMA (113)", 113), ("no barriers (114)", 114), ("same-KiB DMA (118)", 118), ("mfma only (117)", 117)]
res = {t: [] for t, _ in cands}
for t, v in cands:
    bu.prewarm(lambda v=v: host.hgemm_variant(14, 0, 1, 64, v, a, b, c, 1, 2048), 0.3)
for r in range(3):
    for t, v in cands:
        fn = lambda v=v: host.hgemm_variant(14, 0, 1, 64, v, a, b, c, 1, 2048)
        bu.prewarm(fn, 0.15)
        res[t].append(bu.time_region_events(fn, 100))
for t, _ in cands:
    ms = sorted(res[t])[1]
    print("HGABL %-26s %8.4f ms %7.1f TF" % (t, ms, 2.0 * S ** 3 / ms * 1e-9), flush=True)
