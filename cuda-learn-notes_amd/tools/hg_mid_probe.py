"""GPU probe: HGEMM at the mid sizes (where 256x256 tiles leave CUs idle) -- every tile shape the dispatcher can
pick vs the shipped policy and rocBLAS, with a correctness check. python hg_mid_probe.py [sizes...]"""
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
sizes = [int(x) for x in sys.argv[1:]] or [1024, 1536, 2048, 2560, 3072, 3584, 4096, 6144]
#        tag, kind, layout, tile, bk, stages
VARS = [("pp256 split", 8, 0, 1, 64, 4), ("pp256 unsplit", 11, 0, 1, 64, 1), ("pp192 unsplit", 11, 0, 1, 64, 2),
        ("pp192 unsplit TN", 11, 1, 1, 64, 2), ("pp256 dma-in-mma", 13, 0, 1, 64, 4), ("pp256 dma-in-mma TN", 13, 1, 1, 64, 4),
        ("pp192 dma-in-mma", 13, 0, 1, 64, 2), ("pp192 dma-in-mma TN", 13, 1, 1, 64, 2), ("pp256 split TN", 8, 1, 1, 64, 4),
        ("ring 128x128 s3", 0, 0, 0, 64, 3), ("ring 64x128 s3", 0, 0, 6, 64, 3), ("ring 256x128 s2", 0, 0, 2, 64, 2),
        ("ring 128x256 s2", 0, 0, 3, 64, 2), ("ring 128x128 w8 s3", 0, 0, 5, 64, 3)]
for S in sizes:
    torch.manual_seed(S)
    a = torch.randn(S, S, dtype=torch.half, device=dev)
    b = torch.randn(S, S, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    c = torch.zeros(S, S, dtype=torch.half, device=dev)
    ref = (a[:256].float() @ b.float())
    fl = bu.hgemm_flops(S, S, S)
    stride = bu.make_block_swizzle_stride(S, S)
    shipped = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem
    cands = [("rocblas NN", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)), ("rocblas TN", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c)),
             ("shipped " + pkg.manifest.describe(shipped.__name__, (S, S, S), 2)[:28], lambda: shipped(a, b, c, 2, True, stride))]
    for tag, kind, lay, tile, bk, st in VARS:
        fn = lambda kind=kind, lay=lay, tile=tile, bk=bk, st=st: host.hgemm_variant(kind, lay, tile, bk, st, a, bt if lay else b, c, 1, stride)
        try:
            c.zero_()
            fn()
            torch.cuda.synchronize()
            err = (c[:256].float() - ref).abs().max().item()
            print("CHK S=%d %-20s max|err| %.4f %s" % (S, tag, err, "OK" if err < 0.51 else "BAD"), flush=True)
            cands.append((tag, fn))
        except RuntimeError as e:
            print("CHK S=%d %-20s n/a (%s)" % (S, tag, str(e)[:50]), flush=True)
    for tag, fn in cands:
        bu.prewarm(fn, 0.15)
    res = {t: [] for t, _ in cands}
    for rnd in range(3):
        for tag, fn in cands:
            res[tag].append(bu.time_region_events(fn, 30 if S <= 2048 else 12))
    for tag, _ in cands:
        ms = min(res[tag])
        print("HG S=%d %-38s %8.4f ms %7.1f TF   rounds %s" % (S, tag, ms, fl / ms * 1e-9, " ".join("%.4f" % r for r in res[tag])), flush=True)
