import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host
dev = torch.device("cuda:0")
hg = pkg.hgemm_lib(); hg.init_cublas_handle()
for S in (4096, 8192, 2048):
    a = torch.randn(S, S, dtype=torch.half, device=dev); b = torch.randn(S, S, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b); c = torch.zeros(S, S, dtype=torch.half, device=dev)
    fl = bu.hgemm_flops(S, S, S); stride = bu.make_block_swizzle_stride(S, S)
    cands = [("rocblas_nn", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)), ("rocblas_tn", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c))]
    for lay, bb in ((0, b), (1, bt)):
        cands.append(("ring L%d" % lay, lambda lay=lay, bb=bb: host.hgemm_variant(0, lay, 1, 64, 2, a, bb, c, 1, stride)))
        cands.append(("pp   L%d" % lay, lambda lay=lay, bb=bb: host.hgemm_variant(3, lay, 1, 64, 2, a, bb, c, 1, stride)))
        cands.append(("pp   L%d noswz" % lay, lambda lay=lay, bb=bb: host.hgemm_variant(3, lay, 1, 64, 2, a, bb, c, 0, 1)))
    for rnd in range(3):  # interleaved rounds (A/B within one process)
        for tag, fn in cands:
            ms, mn, _ = bu.time_call_events(fn, 2, 10)
            print("S=%d r%d %-16s %8.4f ms %8.1f TF (best %.1f)" % (S, rnd, tag, ms, fl / ms * 1e-9, fl / mn * 1e-9), flush=True)
