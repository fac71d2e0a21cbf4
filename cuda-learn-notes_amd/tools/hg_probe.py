"""GPU probe for the HGEMM kernels: interleaved A/B of explicit variants vs rocBLAS with a correctness check.
python hg_probe.py [sizes...]      variant spec = (tag, kind, layout, stages)"""
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
VARS = [("T128 bk64 s2 NN", 0, 0, 0, 64, 2), ("T128 bk64 s2 TN", 0, 1, 0, 64, 2), ("T128 bk32 s3 NN", 0, 0, 0, 32, 3),
        ("T128W8 bk64 s2 NN", 0, 0, 5, 64, 2), ("T128W8 bk64 s2 TN", 0, 1, 5, 64, 2), ("T128W8 bk32 s4 NN", 0, 0, 5, 32, 4),
        ("T128W8 bk64 s3 NN", 0, 0, 5, 64, 3),
        ("T64x128 bk64 s2 NN", 0, 0, 6, 64, 2), ("T64x128 bk64 s3 NN", 0, 0, 6, 64, 3), ("T64x128 bk64 s3 TN", 0, 1, 6, 64, 3),
        ("T256x128 bk64 s2 NN", 0, 0, 2, 64, 2), ("pp16 split NN", 8, 0, 1, 64, 4)]
for S in sizes:
    torch.manual_seed(S)
    a = torch.randn(S, S, dtype=torch.half, device=dev)
    b = torch.randn(S, S, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    c = torch.zeros(S, S, dtype=torch.half, device=dev)
    ref = (a[:512].float() @ b.float())
    fl = bu.hgemm_flops(S, S, S)
    stride = bu.make_block_swizzle_stride(S, S)
    for tag, kind, lay, tile, bk, st in VARS:
        if "abl" in tag or "nostore" in tag:
            continue
        c.zero_()
        host.hgemm_variant(kind, lay, tile, bk, st, a, bt if lay else b, c, 1, stride)
        torch.cuda.synchronize()
        err = (c[:512].float() - ref).abs().max().item()
        c2 = torch.zeros_like(c)
        host.hgemm_variant(kind, lay, tile, bk, st, a, bt if lay else b, c2, 0, 1)
        same = torch.equal(c, c2)
        print("CHK S=%d %-18s max|err| %.4f (|C|max %.1f) swz-invariant %s %s" % (
            S, tag, err, ref.abs().max().item(), same, "OK" if err < 0.51 and same else "BAD"), flush=True)
    cands = [("rocblas NN", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)),
             ("rocblas TN", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c))]
    for tag, kind, lay, tile, bk, st in VARS:
        cands.append((tag, lambda kind=kind, lay=lay, tile=tile, bk=bk, st=st: host.hgemm_variant(kind, lay, tile, bk, st, a, bt if lay else b, c, 1, stride)))
    for rnd in range(3):
        for tag, fn in cands:
            ms, mn, _ = bu.time_call_events(fn, 3, 12)
            print("HG S=%d r%d %-22s %8.4f ms %7.1f TF (best %7.1f)" % (S, rnd, tag, ms, fl / ms * 1e-9, fl / mn * 1e-9), flush=True)
