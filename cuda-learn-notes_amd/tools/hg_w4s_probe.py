"""GPU probe of the ring-of-slots one-wave-per-SIMD HGEMM (csrc/hgemm_w4s.cuh): S = 2..5 against the stages = 2 kernel,
NN and TN, 4096^3 and 8192^3 (+ an odd multiple-of-64 K), TFLOPS and bit-identity.   python hg_w4s_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host  # noqa: E402

dev = torch.device("cuda:0")
hg = pkg.hgemm_lib()
nn = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem
tn = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4
for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (4096, 4096, 4160), (2048, 4096, 640)):
    torch.manual_seed(1)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    ref, c = torch.zeros(M, N, dtype=torch.half, device=dev), torch.zeros(M, N, dtype=torch.half, device=dev)
    stride = bu.make_block_swizzle_stride(N, K)
    fl = bu.hgemm_flops(M, N, K)
    for layout, fn, bb in ((0, nn, b), (1, tn, bt)):
        call = lambda: fn(a, bb, ref, 2, True, stride)
        bu.prewarm(call, 0.3)
        ms = bu.time_region_events(call, 100 if M <= 4096 else 30)
        print("W4S %-20s %s stages=2 (hgemm_w4)   %8.4f ms %7.1f TF" % ((M, N, K), "TN" if layout else "NN", ms, fl / ms * 1e-9), flush=True)
        for S in (2, 3, 4, 5):
            c.zero_()
            call = lambda: host.hgemm_variant(16, layout, 0, 32, S, a, bb, c, 1, stride)
            try:
                bu.prewarm(call, 0.3)
            except RuntimeError as e:
                print("W4S %-20s %s ring S=%d: %s" % ((M, N, K), "TN" if layout else "NN", S, e))
                continue
            ms = bu.time_region_events(call, 100 if M <= 4096 else 30)
            torch.cuda.synchronize()
            print("W4S %-20s %s ring of %d slots        %8.4f ms %7.1f TF  bit-identical to stages=2: %s" %
                  ((M, N, K), "TN" if layout else "NN", S, ms, fl / ms * 1e-9, bool(torch.equal(c, ref))), flush=True)
    del a, b, bt, ref, c
