"""GPU probe: HGEMM at the launch-bound sizes (config C2 = 1024^3 and neighbours): kernel time by hipGraph replay of
every `stages` value of the top rung, the ring instantiations behind it, and rocBLAS. python hg_small_probe.py [sizes]"""
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
for S in [int(x) for x in sys.argv[1:]] or [512, 1024, 1536, 2048]:
    torch.manual_seed(S)
    a = torch.randn(S, S, dtype=torch.half, device=dev)
    b = torch.randn(S, S, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    c = torch.zeros(S, S, dtype=torch.half, device=dev)
    fl = bu.hgemm_flops(S, S, S)
    stride = bu.make_block_swizzle_stride(S, S)
    top = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem
    cands = [("rocblas NN", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)), ("rocblas TN", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c))]
    cands.append(("1-stage rung (config C2 name) mma2x4_warp4x4", lambda: hg.hgemm_mma_m16n8k16_mma2x4_warp4x4(a, b, c)))
    for st in (2, 3, 4, 5):
        cands.append(("top rung stages=%d %s" % (st, pkg.manifest.describe(top.__name__, (S, S, S), st)[:30]), lambda st=st: top(a, b, c, st, True, stride)))
    #            tag, tile, bk, stages  (ring_exact: tile 0 = 128x128, 6 = 64x128)
    for tile, tname in ((6, "64x128"), (7, "64x64 w4"), (8, "64x64 w2"), (0, "128x128")):
        for bk, st in ((64, 2), (64, 3), (64, 5)):
            fn = lambda tile=tile, bk=bk, st=st: host.hgemm_variant(0, 0, tile, bk, st, a, b, c, 1, stride)
            try:
                c.zero_(); fn(); torch.cuda.synchronize()
                err = (c[:64].float() - (a[:64].float() @ b.float())).abs().max().item()
                if err > 0.51:
                    print("HS S=%d ring %s bk%d s%d BAD err %.3f" % (S, tname, bk, st, err), flush=True)
                    continue
                cands.append(("ring %s bk%d s%d" % (tname, bk, st), fn))
            except RuntimeError:
                pass
    for tag, fn in cands:
        try:
            ms, best = bu.time_call_graph(fn, 20, 5)
            print("HS S=%d %-52s %7.2f us %7.1f TF (best %7.1f)" % (S, tag, ms * 1e3, fl / ms * 1e-9, fl / best * 1e-9), flush=True)
        except Exception as e:
            print("HS", S, tag, "ERR", str(e)[:80], flush=True)
