"""Round-5 probe: the persistent tile walk of the one-wave-per-SIMD HGEMM (csrc/hgemm_w4.cuh EPI 7) against one workgroup per tile, on launches of more
256 x 256 tiles than CUs. The persistent form lives in the probe library (cln_hgemm_variant kind 19); `sum` is an integer checksum of C: equal sums = bit-identical results.
(The committed log was taken while the form was still switchable in the product through $CLN_AMD_W4_PERSIST = 0 / 1, one process each.)
python hg_persist_probe.py [sizes]"""
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
tag = os.environ.get("CLN_AMD_W4_PERSIST", "default")
sizes = [int(x) for x in sys.argv[1:]] or [4096, 5120, 7680, 8192, 8960, 10240, 10752, 12288, 16384]
for S_ in sizes:
    M = N = K = S_
    torch.manual_seed(S_)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    rows = torch.tensor([0, M // 2 + 7, M - 300, M - 1])
    ref = a[rows].float() @ b.float()
    stride = bu.make_block_swizzle_stride(N, K)
    fl = bu.hgemm_flops(M, N, K)
    for name, fn in (("NN", lambda: nn(a, b, c, 2, True, stride)), ("TN", lambda: tn(a, bt, c, 2, True, stride)),
                     ("NN persistent", lambda: host.hgemm_variant(19, 0, 0, 64, 2, a, b, c, 1, stride)),
                     ("TN persistent", lambda: host.hgemm_variant(19, 1, 0, 64, 2, a, bt, c, 1, stride))):
        if "persistent" in name and (M % 256 or (M // 256) ** 2 <= 256):
            continue
        c.zero_()
        fn()
        torch.cuda.synchronize()
        err = (c[rows].float() - ref).abs().max().item()
        chk = int(c.view(torch.int16).to(torch.int64).sum().item())
        bu.prewarm(fn, 0.3)
        ms = min(bu.time_region_events(fn, 20 if S_ <= 8192 else 8) for _ in range(3))
        print("PERSIST=%-7s %5d^3 %-13s %9.2f us %7.1f TF  err %.3f %s sum %d | %s" %
              (tag, S_, name, ms * 1e3, fl / ms * 1e-9, err, "ok" if err < 1e-2 * K ** 0.5 + 0.6 else "WRONG", chk,
               pkg.manifest.describe(bu.HEADLINE_HGEMM_NAME, (M, N, K), 2)[:40]), flush=True)
    del a, b, bt, c
