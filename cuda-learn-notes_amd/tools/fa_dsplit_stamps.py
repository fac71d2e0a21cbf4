"""s_memtime stamps of the d-split attention kernel (probe variants 242 / 243): per wave, cycles between the phase
boundaries of KV tile 16 in workgroup 0.   python fa_dsplit_stamps.py [variant]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import host
dev = torch.device("cuda:0")
B, H, N, D = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,32,4096,512").split(",")]
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
o = torch.zeros_like(q)
names = ["QK loop", "S write+wait", "barrier1", "S read+V issue", "softmax", "PV loop", "vmcnt0", "barrier2"]
for var in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "242,243").split(",")]:
    for rep in range(2):
        host.fa2_variant((4, 0, 15, var), q, k, v, o)
        torch.cuda.synchronize()
    st = o.view(torch.int64).flatten()[:64].cpu().view(8, 8)
    t0 = int(st[:, 0].min())
    print("variant", var, "(s_memtime ticks = 100 MHz constant clock on this part?)")
    print("wave  start " + " ".join("%14s" % n for n in names[:-1] + ["barrier2"]))
    for w in range(8):
        r = st[w].tolist()
        d = [r[i + 1] - r[i] for i in range(7)]
        print("%4d %6d " % (w, r[0] - t0) + " ".join("%14d" % x for x in d), "| total", r[7] - r[0])
