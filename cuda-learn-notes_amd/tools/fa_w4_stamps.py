"""s_memtime stamps of the one-wave-per-SIMD attention kernel (probe variants with ABL & 32): per wave, ticks between
the phase boundaries of iteration 16 in workgroup 0.   python fa_w4_stamps.py [abl,...] [B,H,N,D]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import host
dev = torch.device("cuda:0")
B, H, N, D = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,8,2048,64").split(",")]
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
o = torch.zeros_like(q)
names = ["vmcnt wait", "barrier", "PV 1st half", "PV 2nd half", "QK 1st half", "QK 2nd half"]
for var in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "632,633").split(",")]:
    for rep in range(3):
        host.fa2_variant((4, 0, 0, var), q, k, v, o)
        torch.cuda.synchronize()
    st = o.view(torch.int64).flatten()[:32].cpu().view(4, 8)
    t0 = int(st[:, 0].min())
    print("variant", var, (B, H, N, D))
    print("wave  start " + " ".join("%12s" % n for n in names))
    for w in range(4):
        r = st[w].tolist()
        d = [r[i + 1] - r[i] for i in range(6)]
        print("%4d %6d " % (w, r[0] - t0) + " ".join("%12d" % x for x in d), "| total", r[6] - r[0])
