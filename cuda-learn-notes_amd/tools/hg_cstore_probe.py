"""GPU probe: the C stores of the one-wave-per-SIMD HGEMM kernel as plain / non-temporal / write-through stores (probe variants 26 / 203 / 204
of kind 14), interleaved rounds of event-timed back-to-back launches at 4096^3 and 8192^3, NN and TN.
  python hg_cstore_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host  # noqa: E402

dev = torch.device("cuda:0")
for S in (4096, 8192):
    torch.manual_seed(0)
    a = torch.randn(S, S, dtype=torch.half, device=dev)
    b = torch.randn(S, S, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    c = torch.zeros(S, S, dtype=torch.half, device=dev)
    ref = None
    for lay in (0, 1):
        cands = [(tag, (lambda v=v: host.hgemm_variant(14, lay, 1, 64, v, a, bt if lay else b, c, 1, 2048))) for tag, v in
                 (("plain", 26), ("nt", 203), ("sc0 sc1", 204))]
        for tag, fn in cands:
            c.zero_()
            fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = c.clone()
            assert torch.equal(c, ref), tag
        times = {t: [] for t, _ in cands}
        bu.prewarm(cands[0][1], 0.5)
        for r in range(4):
            for tag, fn in cands:
                bu.prewarm(fn, 0.15)
                times[tag].append(bu.time_region_events(fn, 100 if S == 4096 else 20))
        base = sorted(times["plain"])[1]
        for tag, _ in cands:
            ms = sorted(times[tag])[1]
            print("CST %d %s %-8s %8.4f ms %7.1f TF %+5.1f %%  rounds %s" % (S, "TN" if lay else "NN", tag, ms, 2.0 * S ** 3 / ms * 1e-9, (base / ms - 1) * 100,
                                                                      " ".join("%.4f" % x for x in times[tag])), flush=True)
