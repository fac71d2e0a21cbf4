import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import host
dev = torch.device("cuda:0")
M = N = 4096; K = 8192
a = torch.randn(M, K, dtype=torch.half, device=dev); b = torch.randn(K, N, dtype=torch.half, device=dev)
c = torch.zeros(M, N, dtype=torch.half, device=dev)
hg = pkg.hgemm_lib(); hg.init_cublas_handle()
for rep in range(4):
    host.hgemm_variant(6, 0, 1, 64, 4, a, b, c, 1, 2048)
    for bits in (1, 2, 3, 7):
        host.hgemm_variant(7, 0, 1, 64, bits, a, b, c, 1, 2048)
    host.hgemm_variant(8, 0, 1, 64, 4, a, b, c, 1, 2048)
    hg.hgemm_cublas_tensor_op_nn(a, b, c)
    hg.hgemm_cublas_tensor_op_tn(a, b, c)
torch.cuda.synchronize()
