"""GPU probe: small M x N with a LONG K -- is the latency-bound 64x64 shortcut of best_plan (K <= 2048 only) right to
hand these shapes to the throughput model? Named function (what the policy picks) vs the explicit ring tiles."""
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
from cuda_learn_notes_amd import _loader  # noqa: E402
raw_variant = _loader.load_so("libcln_amd_probe.so").cln_hgemm_variant  # C-ABI directly: the Python wrapper costs as much as a 10 us kernel
raw_nn = _loader.symbol(nn.__name__)
TILES = {7: "ring 64x64", 6: "ring 64x128", 0: "ring 128x128", 3: "ring 128x256", 105: "w4 160x160", 102: "w4 192x192", 103: "w4 128x256"}
SHAPES = ((1024, 1024, 4096), (1024, 1024, 8192), (1536, 1536, 4096), (1536, 1536, 8192), (1280, 1280, 4096), (768, 768, 8192),
          (1024, 2048, 4096), (512, 512, 16384), (1024, 1024, 2048), (1536, 1536, 2048), (1600, 1600, 16384), (1600, 1600, 4096), (1920, 1920, 4096),
          (1280, 1280, 1024), (1536, 1536, 1024), (1024, 1024, 1024), (1152, 1152, 4096), (1536, 1024, 4096))
for (M, N, K) in SHAPES:
    torch.manual_seed(1)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    fl = bu.hgemm_flops(M, N, K)
    rows = []
    ap, bp, cp, st = a.data_ptr(), b.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream
    stride = bu.make_block_swizzle_stride(N, K)
    call = lambda: raw_nn(ap, bp, cp, M, N, K, 2, 1, stride, st)
    bu.prewarm(call, 0.1)
    ms = bu.time_region_events(call, 200)
    rows.append(("policy: " + pkg.manifest.describe(nn.__name__, (M, N, K), 2)[:40], ms))
    for tile, tag in TILES.items():
        if tile >= 100:
            call = lambda: raw_variant(15, 0, tile - 100, 64, 2, ap, bp, cp, M, N, K, 1, stride, st)
        else:
            call = lambda: raw_variant(0, 0, tile, 64, 2, ap, bp, cp, M, N, K, 1, stride, st)
        if call() != 0:
            continue
        bu.prewarm(call, 0.1)
        rows.append((tag, bu.time_region_events(call, 200)))
    best = min(r[1] for r in rows)
    for tag, ms in rows:
        print("SMALLK %-20s %-50s %8.2f us %7.1f TF%s" % ((M, N, K), tag, ms * 1e3, fl / ms * 1e-9, "  <- best" if ms == best else ""), flush=True)
