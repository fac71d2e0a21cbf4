"""GPU probe: tail split (csrc/hgemm_splitk.cuh launch_w4_tail_split) at the sizes of the reference sweep whose count of 256 x 256 tiles is just past a
whole number of rounds of 256: the last `rows` tile rows go to split-K x S, the rest to the single-pass kernel.  python hg_tail_probe.py [sizes]"""
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
nn = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem
fixed256 = hg.hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem  # the 256 x 256 name: hgemm_w4<256x256> at stages = 2
sizes = [int(x) for x in sys.argv[1:]] or [4352, 4608, 4864, 5120, 5888, 6400, 6656, 7168, 7424, 8448, 8704, 9216, 9984, 11008]
for S_ in sizes:
    M = N = K = S_
    torch.manual_seed(S_)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    ref_rows = torch.tensor([0, M // 2, M - 300, M - 1])
    ref = a[ref_rows].float() @ b.float()
    fl = bu.hgemm_flops(M, N, K)
    stride = bu.make_block_swizzle_stride(N, K)
    tiles_n = N // 256
    cands = [("rocblas NN", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)), ("rocblas TN", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c)),
             ("plan", lambda: nn(a, b, c, 2, True, stride)), ("w4 256x256", lambda: fixed256(a, b, c, 2, True, stride))]
    for rows in range(1, M // 256):
        tiles_a = (M // 256 - rows) * tiles_n
        rounds = -(-tiles_a // 256)
        if tiles_a < 200 or rounds * 256 - tiles_a > 40 or rows * tiles_n > 200:  # region A must (nearly) fill whole rounds
            continue
        for S in (2, 3, 4, 5, 6, 8):
            kl = K // S
            if K % (64 * S) or kl < (448 if (kl // 64) & 1 else 384) or rows * tiles_n * S > 320:
                continue
            cands.append(("tail rows=%d S=%d" % (rows, S), lambda rows=rows, S=S: host.hgemm_variant(18, 0, rows, 64, S, a, b, c, 1, stride)))
    res = {}
    for tag, fn in cands:
        c.zero_()
        try:
            fn()
        except RuntimeError as e:
            print("TAIL %d %-20s n/a %s" % (S_, tag, str(e)[:50]), flush=True)
            continue
        torch.cuda.synchronize()
        err = (c[ref_rows].float() - ref).abs().max().item()
        if not err < 1e-2 * K ** 0.5 + 0.6:
            print("TAIL %d %-20s WRONG %.3f" % (S_, tag, err), flush=True)
            continue
        bu.prewarm(fn, 0.05)
        res[tag] = fn
    t = {}
    for rnd in range(2):
        for tag, fn in res.items():
            t[tag] = min(t.get(tag, 1e9), bu.time_region_events(fn, 12))
    tf = {k: fl / v * 1e-9 for k, v in t.items()}
    tails = sorted((k for k in tf if k.startswith("tail")), key=lambda k: -tf[k])
    print("TAIL %5d^3 tiles %4d (%.2f rounds) rocBLAS NN %6.1f TN %6.1f | plan %6.1f (%s) w4-256 %6.1f | %s" % (
        S_, tiles_n * tiles_n, tiles_n * tiles_n / 256.0, tf["rocblas NN"], tf["rocblas TN"], tf["plan"],
        pkg.manifest.describe(nn.__name__, (M, N, K), 2)[:18], tf.get("w4 256x256", float("nan")),
        "  ".join("%s %6.1f" % (k[5:], tf[k]) for k in tails[:4]) or "no tail candidate"), flush=True)
    del a, b, bt, c
