"""GPU probe: the one-wave-per-SIMD HGEMM on 192-row / 192-column tiles (kind 15) vs its 256x256 form, the shipped
dispatcher and rocBLAS at sizes whose 256x256 tiling leaves CUs idle. python hg_w4_shapes_probe.py [sizes...]"""
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
sizes = [int(x) for x in sys.argv[1:]] or [1536, 2304, 3072, 4608, 6144, 7680]
SHAPES = {0: (192, 256), 1: (256, 192), 2: (192, 192), 3: (128, 256), 4: (256, 128), 5: (160, 160)}
ONLY = [int(x) for x in os.environ.get('W4_SHAPES', '0,1,2,3,4,5').split(',')]
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
    try:
        what = pkg.manifest.describe(shipped.__name__, (S, S, S), 2)
    except ValueError:
        what = "(unsupported)"
    cands = [("rocblas NN", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)), ("rocblas TN", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c)),
             ("shipped NN " + what[:22], lambda: shipped(a, b, c, 2, True, stride)),
             ("shipped TN", lambda: shipped_tn(a, bt, c, 2, True, stride))]
    if what == "(unsupported)":
        cands = cands[:2]
    var = []
    if S % 256 == 0:
        var += [("w4 256x256 NN", 14, 0, 1, 26), ("w4 256x256 TN", 14, 1, 1, 26)]
    for t, (bm, bn) in SHAPES.items():
        if S % bm == 0 and S % bn == 0 and t in ONLY:
            var += [("w4 %dx%d NN" % (bm, bn), 15, 0, t, 2), ("w4 %dx%d TN" % (bm, bn), 15, 1, t, 2)]
    for tag, kind, lay, tile, st in var:
        fn = lambda kind=kind, lay=lay, tile=tile, st=st: host.hgemm_variant(kind, lay, tile, 64, st, a, bt if lay else b, c, 1, stride)
        try:
            c.zero_()
            fn()
            torch.cuda.synchronize()
            err = (c.float() - ref.float()).abs().max().item()
            print("CHK S=%d %-20s max|err| vs rocBLAS %.4f %s" % (S, tag, err, "OK" if err < 0.51 else "BAD"), flush=True)
            if err < 0.51:
                cands.append((tag, fn))
        except RuntimeError as e:
            print("CHK S=%d %-20s n/a (%s)" % (S, tag, str(e)[:60]), flush=True)
    for tag, fn in cands:
        bu.prewarm(fn, 0.15)
    res = {t: [] for t, _ in cands}
    for rnd in range(3):
        for tag, fn in cands:
            res[tag].append(bu.time_region_events(fn, 30 if S <= 2304 else (16 if S <= 4608 else 6)))
    for tag, _ in cands:
        ms = min(res[tag])
        print("HG S=%d %-34s best %8.4f ms %7.1f TF   rounds %s" % (S, tag, ms, fl / ms * 1e-9, " ".join("%.4f" % r for r in res[tag])), flush=True)
