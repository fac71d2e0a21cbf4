"""GPU check of the hipBLASLt comparison row (csrc/hgemm_vendor_lt.hip): result vs torch fp32 matmul, and event-timed TFLOPS
beside rocBLAS and the shipped kernel at a few squares.  python vendor_lt_probe.py [sizes...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu  # noqa: E402

dev = torch.device("cuda:0")
hg = pkg.hgemm_lib()
lt = pkg.load("hgemm_vendor_lt")
hg.init_cublas_handle()
for S in [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096, 8192]:
    torch.manual_seed(0)
    a = torch.randn(S, S, dtype=torch.half, device=dev)
    b = torch.randn(S, S, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    c = torch.zeros(S, S, dtype=torch.half, device=dev)
    ref = (a[:256].float() @ b.float())
    rows = [("hipblaslt_nn", lambda: lt.cln_hgemm_hipblaslt_nn(a, b, c)), ("hipblaslt_tn", lambda: lt.cln_hgemm_hipblaslt_tn(a, bt, c)),
            ("rocblas_nn", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)), ("rocblas_tn", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c)),
            ("ours_nn", lambda: hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem(a, b, c, 2, True, bu.make_block_swizzle_stride(S, S))),
            ("ours_tn", lambda: hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4(a, bt, c, 2, True, bu.make_block_swizzle_stride(S, S)))]
    for tag, fn in rows:
        c.zero_()
        fn()
        torch.cuda.synchronize()
        err = (c[:256].float() - ref).abs().max().item()
        for _ in range(30):
            fn()
        ms, mn, _ = bu.time_call_events(fn, 5, 50)
        print("LT %5d %-13s max|err| %.3e  %8.4f ms %7.1f TF" % (S, tag, err, ms, 2.0 * S ** 3 / ms * 1e-9), flush=True)
hg.destroy_cublas_handle()
