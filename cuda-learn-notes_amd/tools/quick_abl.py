import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host
dev = torch.device("cuda:0")
M = N = 4096
for rnd in range(2):
    for K in (4096, 8192):
        a = torch.randn(M, K, dtype=torch.half, device=dev); b = torch.randn(K, N, dtype=torch.half, device=dev)
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        cands = [("pp4-nostore", lambda: host.hgemm_variant(6, 0, 1, 64, 4, a, b, c, 1, 2048))]
        for bits, name in ((1, "no-fragreads"), (2, "no-dma"), (3, "no-frag,no-dma"), (7, "mfma only (no barrier)"), (8, "no-setprio"), (4, "no-barrier")):
            cands.append(("abl %d %s" % (bits, name), lambda bits=bits: host.hgemm_variant(7, 0, 1, 64, bits, a, b, c, 1, 2048)))
        for tag, fn in cands:
            ms, mn, _ = bu.time_call_events(fn, 3, 15)
            print("K=%5d r%d %-28s %8.2f us (min %8.2f)  %7.1f TF-equiv" % (K, rnd, tag, ms * 1e3, mn * 1e3, 2.0 * M * N * K / ms * 1e-9), flush=True)
