"""GPU probe: the one-wave-per-SIMD HGEMM (csrc/hgemm_w4.cuh, 128x128 wave tiles) vs the shipped ping-pong kernel and
rocBLAS, NN and TN, with a full-matrix correctness check. python hg_w4_probe.py [sizes...]"""
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
hg.init_cublas_handle()
sizes = [int(x) for x in sys.argv[1:]] or [4096, 8192]
VLIST = [int(x) for x in os.environ.get("W4_VARS", "1,4,5,6,7,8,9,104").split(",")]
VARS = []
for v in VLIST:
    VARS.append(("w4 v%d NN" % v, 14, 0, 1, 64, v))
    VARS.append(("w4 v%d TN" % v, 14, 1, 1, 64, v))
for S in sizes:
    torch.manual_seed(S)
    a = torch.randn(S, S, dtype=torch.half, device=dev)
    b = torch.randn(S, S, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    c = torch.zeros(S, S, dtype=torch.half, device=dev)
    ref = torch.empty_like(c)
    hg.hgemm_cublas_tensor_op_nn(a, b, ref)
    torch.cuda.synchronize()
    fl = bu.hgemm_flops(S, S, S)
    stride = bu.make_block_swizzle_stride(S, S)
    shipped = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem
    shipped_tn = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4
    cands = [("rocblas NN", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)), ("rocblas TN", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c)),
             ("shipped NN", lambda: shipped(a, b, c, 2, True, stride)), ("shipped TN", lambda: shipped_tn(a, bt, c, 2, True, stride))]
    for tag, kind, lay, tile, bk, st in VARS:
        fn = lambda kind=kind, lay=lay, tile=tile, bk=bk, st=st: host.hgemm_variant(kind, lay, tile, bk, st, a, bt if lay else b, c, 1, stride)
        try:
            c.zero_()
            fn()
            torch.cuda.synchronize()
            err = (c.float() - ref.float()).abs().max().item()
            print("CHK S=%d %-20s max|err| vs rocBLAS %.4f %s" % (S, tag, err, "OK" if err < 0.51 else "BAD"), flush=True)
            if err < 0.51 or st >= 100:
                cands.append((tag, fn))
        except RuntimeError as e:
            print("CHK S=%d %-20s n/a (%s)" % (S, tag, str(e)[:50]), flush=True)
    for tag, fn in cands:
        bu.prewarm(fn, 0.2)
    res = {t: [] for t, _ in cands}
    for rnd in range(4):
        for tag, fn in cands:
            res[tag].append(bu.time_region_events(fn, 20 if S <= 4096 else 6))
    for tag, _ in cands:
        ms = min(res[tag])
        av = sum(res[tag]) / len(res[tag])
        print("HG S=%d %-14s best %8.4f ms %7.1f TF  mean %7.1f TF  rounds %s" % (S, tag, ms, fl / ms * 1e-9, fl / av * 1e-9, " ".join("%.4f" % r for r in res[tag])), flush=True)

# K scaling at M = N = 4096: time = intercept (launch ramp + prologue + epilogue) + slope * K-tiles
KV = int(os.environ.get("W4_KSCALE", "4"))
if KV >= 0:
    M = N = 4096
    for K in (1024, 2048, 4096, 8192, 16384):
        a = torch.randn(M, K, dtype=torch.half, device=dev)
        b = torch.randn(K, N, dtype=torch.half, device=dev)
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        stride = bu.make_block_swizzle_stride(N, K)
        shipped = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem
        for tag, fn in (("shipped NN", lambda: shipped(a, b, c, 2, True, stride)),
                        ("w4 v%d NN" % KV, lambda: host.hgemm_variant(14, 0, 1, 64, KV, a, b, c, 1, stride)),
                        ("w4 v%d NN no-store" % KV, lambda: host.hgemm_variant(14, 0, 1, 64, 104, a, b, c, 1, stride))):
            bu.prewarm(fn, 0.2)
            ms = min(bu.time_region_events(fn, 20) for _ in range(3))
            print("KS M=N=4096 K=%5d %-20s %8.4f ms %7.1f TF  (%.3f us per K tile)" % (K, tag, ms, 2.0 * M * N * K / ms * 1e-9, ms * 1e3 / (K / 64)), flush=True)
