"""GPU probe: HGEMM on rectangular shapes (tall / wide / long-K / short-K): the shipped policy (stages 2, NN and TN names) against
rocBLAS NN / TN and hipBLASLt, with a correctness check on the first 128 rows.  Finds holes in csrc/hgemm.hip best_plan, which was
fitted on squares.  python hg_rect_probe.py [squares]"""
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
hg.init_cublas_handle()
SHAPES = [(16384, 1024, 4096), (1024, 16384, 4096), (8192, 2048, 8192), (2048, 8192, 2048), (4096, 11008, 4096), (4096, 4096, 11008),
          (8192, 28672, 8192), (8192, 8192, 28672), (512, 8192, 8192), (256, 4096, 4096), (128, 8192, 8192), (32768, 512, 512),
          (4096, 14336, 4096), (16384, 16384, 1024), (16384, 16384, 512), (8192, 8192, 256), (8192, 1024, 1024), (1024, 8192, 1024),
          (2048, 2048, 8192), (1024, 1024, 16384), (4096, 12288, 4096), (4096, 4096, 16384), (16384, 4096, 4096), (3072, 9216, 3072),
          (5120, 13824, 5120), (640, 5120, 5120), (8192, 8192, 8192 + 64), (8192 + 256, 8192, 8192), (4096, 4096 + 128, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "squares":  # the reference's default sweep: M = N = K = 256 ... 12800 step 256 (hgemm.py:22-23, :277-281, :308)
    SHAPES = [(s, s, s) for s in range(256, 12800 + 256, 256)]
nn = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem
tn = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4
for (M, N, K) in SHAPES:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    ref = a[:128].float() @ b.float()
    fl = bu.hgemm_flops(M, N, K)
    stride = bu.make_block_swizzle_stride(N, K)
    cands = [("rocblas NN", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)), ("rocblas TN", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c)),
             ("ours NN", lambda: nn(a, b, c, 2, True, stride)), ("ours TN", lambda: tn(a, bt, c, 2, True, stride))]
    out = {}
    for tag, fn in cands:
        c.zero_()
        fn()
        torch.cuda.synchronize()
        err = (c[:128].float() - ref).abs().max().item()
        assert err < 1e-2 * K ** 0.5 + 0.6, (tag, M, N, K, err)
        bu.prewarm(fn, 0.1)
    rep = max(6, min(200, int(2e12 / fl * 20)))
    for rnd in range(2):
        for tag, fn in cands:
            ms = bu.time_region_events(fn, rep)
            out[tag] = min(out.get(tag, 1e9), ms)
    tfs = {t: fl / out[t] * 1e-9 for t in out}
    print("RECT %6d x %6d x %6d  rocBLAS NN %7.1f TN %7.1f | ours NN %7.1f (%4.0f%%) TN %7.1f (%4.0f%%)  %s"
          % (M, N, K, tfs["rocblas NN"], tfs["rocblas TN"], tfs["ours NN"], 100 * tfs["ours NN"] / tfs["rocblas NN"], tfs["ours TN"],
             100 * tfs["ours TN"] / tfs["rocblas TN"], pkg.manifest.describe(nn.__name__, (M, N, K), 2)[:40]), flush=True)
    del a, b, bt, c, ref
