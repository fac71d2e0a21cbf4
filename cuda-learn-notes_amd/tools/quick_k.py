import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host
dev = torch.device("cuda:0")
hg = pkg.hgemm_lib(); hg.init_cublas_handle()
M = N = 4096
for rnd in range(2):
    for K in (4096, 8192):
        a = torch.randn(M, K, dtype=torch.half, device=dev); b = torch.randn(K, N, dtype=torch.half, device=dev)
        c = torch.zeros(M, N, dtype=torch.half, device=dev); bt = bu.as_col_major(b)
        for tag, fn in (("rocblas_nn", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)),
                        ("ring", lambda: host.hgemm_variant(0, 0, 1, 64, 2, a, b, c, 1, 2048)),
                        ("pp", lambda: host.hgemm_variant(3, 0, 1, 64, 2, a, b, c, 1, 2048)),
                        ("pp-nostore", lambda: host.hgemm_variant(4, 0, 1, 64, 2, a, b, c, 1, 2048)),
                        ("pp-ldsepi-8", lambda: host.hgemm_variant(5, 0, 1, 64, 8, a, b, c, 1, 2048)),
                        ("pp-ldsepi-4", lambda: host.hgemm_variant(5, 0, 1, 64, 4, a, b, c, 1, 2048)),
                        ("pp4-nostore", lambda: host.hgemm_variant(6, 0, 1, 64, 4, a, b, c, 1, 2048)),
                        ("pp4-split", lambda: host.hgemm_variant(8, 0, 1, 64, 4, a, b, c, 1, 2048)),
                        ("pp4-split-nostore", lambda: host.hgemm_variant(8, 0, 1, 64, 1, a, b, c, 1, 2048)),
                        ("pp32", lambda: host.hgemm_variant(9, 0, 1, 32, 4, a, b, c, 1, 2048)),
                        ("pp32-nostore", lambda: host.hgemm_variant(9, 0, 1, 32, 1, a, b, c, 1, 2048)),
                        ("rocblas_tn", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c)),
                        ("pp32 TN", lambda: host.hgemm_variant(9, 1, 1, 32, 4, a, bt, c, 1, 2048)),
                        ("pp4-split TN", lambda: host.hgemm_variant(8, 1, 1, 64, 4, a, bt, c, 1, 2048))):
            ms, mn, _ = bu.time_call_events(fn, 3, 15)
            print("K=%5d r%d %-12s %8.2f us (min %8.2f)  %7.1f TF" % (K, rnd, tag, ms * 1e3, mn * 1e3, 2.0 * M * N * K / ms * 1e-9), flush=True)
