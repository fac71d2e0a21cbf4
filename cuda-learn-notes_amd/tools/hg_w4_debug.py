"""Debug helper: which 16x16 blocks of one 256x256 tile does a w4 variant get wrong? python hg_w4_debug.py VAR K [layout]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host
dev = torch.device("cuda:0")
var = int(sys.argv[1]); K = int(sys.argv[2]); lay = int(sys.argv[3]) if len(sys.argv) > 3 else 1
M = N = 256
torch.manual_seed(1)
a = torch.randn(M, K, dtype=torch.half, device=dev)
b = torch.randn(K, N, dtype=torch.half, device=dev)
bt = bu.as_col_major(b)
c = torch.zeros(M, N, dtype=torch.half, device=dev)
host.hgemm_variant(14, lay, 1, 64, var, a, bt if lay else b, c, 1, 256)
torch.cuda.synchronize()
ref = a.float() @ b.float()
err = (c.float() - ref).abs().view(16, 16, 16, 16).amax(dim=(1, 3))
print("var", var, "K", K, "max err", err.max().item())
print((err > 0.5).int().cpu().numpy())
# which K tiles contribute wrongly: per-K-tile partial products
for t in range(K // 64):
    a2 = torch.zeros_like(a); a2[:, t*64:(t+1)*64] = a[:, t*64:(t+1)*64]
    c.zero_()
    host.hgemm_variant(14, lay, 1, 64, var, a2, bt if lay else b, c, 1, 256)
    torch.cuda.synchronize()
    r2 = a2.float() @ b.float()
    e = (c.float() - r2).abs().view(16, 16, 16, 16).amax(dim=(1, 3))
    print("only K tile", t, "bad blocks", int((e > 0.5).sum().item()), "max", e.max().item())
