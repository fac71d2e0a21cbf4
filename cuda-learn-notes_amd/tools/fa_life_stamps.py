"""Where a short attention launch spends its time: s_memrealtime stamps (100 MHz) of EVERY wave of the shipped D = 64
ping-pong kernel (probe variant 528 = ABL 128): kernel entry, end of the prologue, start of KV tiles 1 2 3 4 8 and of
the last tile, end of the KV loop, O stores issued / acknowledged.   python fa_life_stamps.py [B,H,N,D]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import host
dev = torch.device("cuda:0")
B, H, N, D = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,8,2048,64").split(",")]
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
o = torch.zeros_like(q)
names = ["entry", "prologue", "tile1", "tile2", "tile3", "tile4", "tile8", "lastT", "loopend", "stored", "acked"]
for rep in range(30):
    host.fa2_variant((8, 0, 0, 528), q, k, v, o)
torch.cuda.synchronize()
for rep in range(3):
    host.fa2_variant((8, 0, 0, 528), q, k, v, o)
    torch.cuda.synchronize()
    st = o.view(torch.int64).view(-1, 32 * D // 4)[:, :12].cpu()  # one row per wave (32 query rows)
    t = st[:, :11].double() * 0.01  # us
    t0 = t[:, 0].min()
    t = t - t0
    xcc = st[:, 11] & 15
    print("shape", (B, H, N, D), "rep", rep, "waves", t.shape[0], "kernel span %.2f us (first entry -> last ack)" % float(t[:, 10].max()))
    print("%-9s %8s %8s %8s %8s" % ("stamp", "min", "median", "p95", "max"))
    for i, n in enumerate(names):
        c = t[:, i]
        print("%-9s %8.2f %8.2f %8.2f %8.2f" % (n, c.min(), c.median(), c.quantile(0.95), c.max()))
    # per-wave durations
    segs = [("prologue", 0, 1), ("tile0", 1, 2), ("tile1", 2, 3), ("tile2", 3, 4), ("tile3", 4, 5), ("tiles4-7 /4", 5, 6), ("tile8..last /n", 6, 7), ("epilogue", 8, 9), ("store ack", 9, 10), ("whole", 0, 10)]
    T = N // 128
    for n, a, b in segs:
        d = t[:, b] - t[:, a]
        if n.startswith("tiles4"):
            d = d / 4
        if n.startswith("tile8"):
            d = d / max(T - 1 - 8, 1)
        print("  %-16s min %6.2f median %6.2f p95 %6.2f max %6.2f us" % (n, d.min(), d.median(), d.quantile(0.95), d.max()))
    grp = (torch.arange(t.shape[0]) // 4) % 2
    for g in (0, 1):
        m = grp == g
        print("  group %d: entry median %.2f, loop end median %.2f, acked median %.2f max %.2f" % (g, t[m, 0].median(), t[m, 8].median(), t[m, 10].median(), t[m, 10].max()))
    for x in range(8):
        m = xcc == x
        if m.any():
            print("  xcc %d: waves %4d entry median %.2f acked median %.2f max %.2f" % (x, int(m.sum()), t[m, 0].median(), t[m, 10].median(), t[m, 10].max()))
