"""The reference's sgemm sweep (kernels/sgemm/sgemm.py:109-120: M, N in {4096, 8192, 16384}, K in {2048, 4096, 8192}) plus off-sweep shapes through the product's
matrix-core entry point, against rocBLAS sgemm (`sgemm_cublas`) and torch.matmul (hipBLASLt on this image): TFLOPS, fraction of the 157.3 TF f32 matrix peak,
ratio to the better vendor row, and the maximum error of sampled rows against the fp64 product.
    python tools/sgemm_sweep.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from cuda_learn_notes_amd import _loader, bench_utils as bu
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
prod = _loader.symbol("sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages")
cub = _loader.symbol("sgemm_cublas") if _loader.has_symbol("sgemm_cublas") else None
if cub:
    _loader.symbol("init_cublas_handle")()
torch.backends.cuda.matmul.allow_tf32 = False
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
shapes = [(M, N, K) for M in (4096, 8192, 16384) for N in (4096, 8192, 16384) for K in (2048, 4096, 8192)]
if quick:
    shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (16384, 4096, 2048)]
shapes += [(4864, 4864, 4864), (5120, 5120, 5120), (6144, 6144, 6144), (3072, 3072, 3072), (2048, 2048, 2048), (1024, 1024, 1024), (2560, 5120, 4096), (4096, 4096, 256)]
worst = 10.0
for (M, N, K) in shapes:
    if M * N * K > 16384 * 16384 * 4096:
        continue
    a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev); c = torch.zeros(M, N, device=dev)
    ap, bp, cp = a.data_ptr(), b.data_ptr(), c.data_ptr()
    fl = 2.0 * M * N * K

    def t(call):
        bu.prewarm(call, 0.05)
        it = max(3, min(40, int(0.05 / (fl / 140e12)) + 1))
        return fl / bu.time_region_events(call, it) * 1e-9
    roc = t(lambda: cub(ap, bp, cp, M, N, K, st)) if cub else float("nan")
    tor = t(lambda: torch.matmul(a, b, out=c))
    assert prod(ap, bp, cp, M, N, K, 2, 1, 256, st) == 0
    ours = t(lambda: prod(ap, bp, cp, M, N, K, 2, 1, 256, st))
    sel = torch.tensor([0, M // 2 + 1, M - 1], device=dev)
    truth = a[sel].double() @ b.double()
    err = ((c[sel].double() - truth).abs().max() / truth.abs().max()).item()
    best = max(roc, tor) if cub else tor
    in_sweep = M >= 4096 and N >= 4096 and K >= 2048 and M in (4096, 8192, 16384) and N in (4096, 8192, 16384)
    if in_sweep:
        worst = min(worst, ours / best)
    print("SGSWEEP %6d %6d %6d  rocBLAS %6.1f  torch.matmul %6.1f | ours %6.1f TF (%.3f of 157.3)  x best vendor %.3f  rel err %.1e%s"
          % (M, N, K, roc, tor, ours, ours / 157.3, ours / best, err, "" if in_sweep else "  (off the reference's sweep)"), flush=True)
    del a, b, c
print("SGSWEEP worst ratio to the better vendor row on the reference's sweep: %.3f" % worst)
