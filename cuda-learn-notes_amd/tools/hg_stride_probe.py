"""N-band (block swizzle stride) sweep of the 256x256 HGEMM kernels vs rocBLAS. HG_STRIDE_W4=1: the one-wave-per-SIMD kernel (kind 14, schedule 26) instead of the ping-pong kernel."""
import os, sys, torch
KIND, ST = ((14, 26) if os.environ.get("HG_STRIDE_W4") else (8, 4))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host
dev = torch.device("cuda:0")
hg = pkg.hgemm_lib()
hg.init_cublas_handle()
for S in [int(x) for x in sys.argv[1:]] or [8192, 4096]:
    torch.manual_seed(S)
    a = torch.randn(S, S, dtype=torch.half, device=dev)
    b = torch.randn(S, S, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    c = torch.zeros(S, S, dtype=torch.half, device=dev)
    fl = bu.hgemm_flops(S, S, S)
    cands = [("rocblas NN", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)), ("rocblas TN", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c))]
    for lay in (0, 1):
        for stride in (256, 512, 1024, 2048, 4096, S):
            cands.append(("pp %s stride %d" % ("TN" if lay else "NN", stride),
                          lambda lay=lay, stride=stride: host.hgemm_variant(KIND, lay, 1, 64, ST, a, bt if lay else b, c, 1, stride)))
        cands.append(("pp %s no swizzle" % ("TN" if lay else "NN"), lambda lay=lay: host.hgemm_variant(KIND, lay, 1, 64, ST, a, bt if lay else b, c, 0, 1)))
    for rnd in range(2):
        for tag, fn in cands:
            ms, mn, _ = bu.time_call_events(fn, 3, 12 if S >= 8192 else 30)
            if rnd:
                print("HG S=%d %-24s %8.4f ms %7.1f TF" % (S, tag, ms, fl / ms * 1e-9), flush=True)
