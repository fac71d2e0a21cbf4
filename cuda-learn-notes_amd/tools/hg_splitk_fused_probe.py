"""Round-5 probe: the PRODUCTION best-dispatch name on the split-K / tail-split shapes of profiles/r04_hgemm_splitk_probe.log and
r04_hgemm_tail_probe_after.log, with the in-kernel fix-up (one launch, csrc/hgemm_w4.cuh EPI 6) taken up to $CLN_AMD_SPLITK_FUSED_MAX_S splits.
The environment variable is read once per process: run this script once per setting (tools/round_evidence.sh does 0 = always two launches,
4 = default, 64 = always one launch) and compare the rows.   python hg_splitk_fused_probe.py [--quick]"""
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
nn = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem
tag = os.environ.get("CLN_AMD_SPLITK_FUSED_MAX_S", "default(4)")
shapes = [(512, 8192, 8192), (256, 4096, 4096), (128, 8192, 8192), (2048, 2048, 8192), (1024, 1024, 16384), (640, 5120, 5120), (1024, 1024, 4096),
          (1024, 1024, 8192), (1536, 1536, 8192), (2048, 2048, 16384), (256, 256, 16384), (512, 512, 8192), (1024, 4096, 8192), (1280, 1280, 8192),
          (768, 768, 12288), (256, 8192, 4096), (128, 4096, 4096), (512, 2048, 16384),
          (4352, 4352, 4352), (4864, 4864, 4864), (5888, 5888, 5888), (7168, 7168, 7168), (8448, 8448, 8448), (9216, 9216, 9216)]
if "--quick" in sys.argv:
    shapes = shapes[:6] + shapes[-6:-3]
for (M, N, K) in shapes:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    rows = torch.tensor(sorted({0, M // 2, max(M - 300, 0), M - 1}))
    ref = a[rows].float() @ b.float()
    stride = bu.make_block_swizzle_stride(N, K)
    fn = lambda: nn(a, b, c, 2, True, stride)
    fn()
    torch.cuda.synchronize()
    err = (c[rows].float() - ref).abs().max().item()
    ok = err < 1e-2 * K ** 0.5 + 0.6
    c2 = torch.zeros_like(c)
    nn(a, b, c2, 2, True, stride)  # a second launch on the self-reset tickets must give the same bits
    torch.cuda.synchronize()
    same = torch.equal(c, c2)
    bu.prewarm(fn, 0.1)
    ms = min(bu.time_region_events(fn, 30) for _ in range(2))
    plan = pkg.manifest.describe(bu.HEADLINE_HGEMM_NAME, (M, N, K), 2)
    print("SKF max_s=%-11s %5d x %5d x %5d  %8.2f us %7.1f TF  err %.3f %s rerun-identical %s | %s" %
          (tag, M, N, K, ms * 1e3, bu.hgemm_flops(M, N, K) / ms * 1e-9, err, "ok" if ok else "WRONG", same, plan[-70:]), flush=True)
    del a, b, c, c2
