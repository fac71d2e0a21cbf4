"""GPU probe: split-K over the one-wave-per-SIMD HGEMM (csrc/hgemm_splitk.cuh) on shapes whose M x N gives few tiles and whose K is
long -- every (tile shape, number of splits) candidate against the shipped policy and rocBLAS NN / TN, with a correctness check.
The table csrc/hgemm.hip splitk_plan is fitted to.  python hg_splitk_probe.py [tn]"""
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
LAYOUT = 1 if "tn" in sys.argv[1:] else 0
SHAPES = [(512, 8192, 8192), (256, 4096, 4096), (128, 8192, 8192), (2048, 2048, 8192), (1024, 1024, 16384), (640, 5120, 5120),
          (1024, 1024, 4096), (1024, 1024, 8192), (1536, 1536, 8192), (2048, 2048, 4096), (2048, 2048, 16384), (256, 256, 16384),
          (512, 512, 8192), (1024, 4096, 8192), (4096, 1024, 8192), (2560, 2560, 8192), (1280, 1280, 8192), (768, 768, 12288),
          (256, 8192, 4096), (8192, 256, 4096), (128, 4096, 4096), (1024, 1024, 2048), (2048, 2048, 2048), (1024, 1024, 1024),
          (1536, 1536, 3072), (3072, 3072, 8192), (2048, 1024, 4096), (512, 2048, 16384)]
TILES = {0: (256, 256), 1: (128, 256), 2: (256, 128), 3: (192, 256), 4: (192, 192), 5: (160, 160)}
nn = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem
tn = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4
for (M, N, K) in SHAPES:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    bb = bt if LAYOUT else b
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    ref = a[:128].float() @ b.float()
    fl = bu.hgemm_flops(M, N, K)
    stride = bu.make_block_swizzle_stride(N, K)
    ours = (lambda: tn(a, bt, c, 2, True, stride)) if LAYOUT else (lambda: nn(a, b, c, 2, True, stride))
    cands = [("rocblas NN", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)), ("rocblas TN", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c)),
             ("plan", ours)]
    for t, (bm, bn) in TILES.items():
        if M % bm or N % bn:
            continue
        tiles = (M // bm) * (N // bn)
        for S in (2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32):
            if tiles * S > 640 or tiles * S < 48 or K % (64 * S):
                continue
            kl = K // S
            if kl < (448 if (kl // 64) & 1 else 384):
                continue
            cands.append(("%dx%d S=%d" % (bm, bn, S), lambda t=t, S=S: host.hgemm_variant(17, LAYOUT, t, 64, S, a, bb, c, 0, 0)))
    rep = max(6, min(100, int(1.5e12 / fl * 20)))
    res = {}
    for tag, fn in cands:
        c.zero_()
        try:
            fn()
        except RuntimeError as e:
            print("SPLITK %d x %d x %d %-16s n/a %s" % (M, N, K, tag, str(e)[:60]), flush=True)
            continue
        torch.cuda.synchronize()
        err = (c[:128].float() - ref).abs().max().item()
        if not err < 1e-2 * K ** 0.5 + 0.6:
            print("SPLITK %d x %d x %d %-16s WRONG max|err| %.3f" % (M, N, K, tag, err), flush=True)
            continue
        bu.prewarm(fn, 0.03)
        res[tag] = min(bu.time_region_events(fn, rep), bu.time_region_events(fn, rep))
    tfs = {t: fl / res[t] * 1e-9 for t in res}
    sk = sorted((t for t in tfs if "S=" in t), key=lambda t: -tfs[t])
    desc = pkg.manifest.describe((tn if LAYOUT else nn).__name__, (M, N, K), 2)[:34]
    print("SPLITK %5d x %5d x %5d %s rocBLAS NN %6.1f TN %6.1f | plan %6.1f (%s) | best split-K: %s" % (
        M, N, K, "TN" if LAYOUT else "NN", tfs["rocblas NN"], tfs["rocblas TN"], tfs["plan"], desc,
        "  ".join("%s %6.1f" % (t, tfs[t]) for t in sk[:5])), flush=True)
    print("SPLITKALL %d %d %d %s" % (M, N, K, " ".join("%s=%.1f" % (t.replace(" ", ""), tfs[t]) for t in tfs)), flush=True)
    del a, b, bt, c, ref
