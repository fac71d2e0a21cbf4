"""One-launch against two-launch split-K on the planner's own (tile, splits) choice (round 6 form of the round-5 probe: the environment knob that
switched the product between the forms is gone -- libcln_amd.so reads one variable, $CLN_AMD_NO_SPLITK -- so both forms are launched through the
TEST-ONLY probe hook: kind 17 = partials + hgemm_splitk_reduce launch, kind 20 = in-kernel fix-up by the last-arriving workgroup of a tile, at ANY
number of splits). Per shape: the production name (what the planner ships), the two probe forms on the same (tile, S), bit-identity of the three,
error of sampled rows against the fp32 product. The product takes the one-launch form at 2 splits only (csrc/hgemm.hip splitk_fused_max_s).
Lines start with SKF.   python hg_splitk_fused_probe.py [--quick]"""
import os
import re
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
TILE_CODE = {(256, 256): 0, (128, 256): 1, (192, 256): 3, (192, 192): 4, (160, 160): 5}
shapes = [(512, 8192, 8192), (256, 4096, 4096), (128, 8192, 8192), (2048, 2048, 8192), (1024, 1024, 16384), (640, 5120, 5120), (1024, 1024, 4096),
          (1024, 1024, 8192), (1536, 1536, 8192), (2048, 2048, 16384), (256, 256, 16384), (512, 512, 8192), (1024, 4096, 8192), (1280, 1280, 8192),
          (768, 768, 12288), (256, 8192, 4096), (128, 4096, 4096), (512, 2048, 16384)]
if "--quick" in sys.argv:
    shapes = shapes[:8]
for (M, N, K) in shapes:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    rows = torch.tensor(sorted({0, M // 2, max(M - 300, 0), M - 1}))
    ref = a[rows].float() @ b.float()
    plan = pkg.manifest.describe(bu.HEADLINE_HGEMM_NAME, (M, N, K), 2)
    m = re.match(r"hgemm_w4<(\d+)x(\d+)x64.*split-K x (\d+)", plan)
    if not m:
        print("SKF %5d x %5d x %5d  not a split-K plan: %s" % (M, N, K, plan[:80]), flush=True)
        continue
    tile, S = TILE_CODE[(int(m.group(1)), int(m.group(2)))], int(m.group(3))
    outs, times = {}, {}
    forms = (("production", lambda c: nn(a, b, c, 2, True, bu.make_block_swizzle_stride(N, K))),
             ("two launches", lambda c: host.hgemm_variant(17, 0, tile, 64, S, a, b, c)), ("one launch", lambda c: host.hgemm_variant(20, 0, tile, 64, S, a, b, c)))
    for name, run in forms:
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        run(c)
        torch.cuda.synchronize()
        first = c.clone()
        run(c)  # a second launch on the self-reset tickets must give the same bits
        torch.cuda.synchronize()
        outs[name] = (c, torch.equal(c, first))
        bu.prewarm(lambda: run(c), 0.1)
        times[name] = min(bu.time_region_events(lambda: run(c), 30) for _ in range(2))
    err = (outs["production"][0][rows].float() - ref).abs().max().item()
    ok = err < 1e-2 * K ** 0.5 + 0.6
    same = all(torch.equal(outs[n][0], outs["production"][0]) and outs[n][1] for n in outs)
    print("SKF %5d x %5d x %5d  tile %sx%s S=%-2d  production %7.2f us %6.1f TF | two launches %7.2f us | one launch %7.2f us (%+5.1f %%)  err %.3f %s  all three bit-identical and rerun-identical: %s  [product form: %s]" %
          (M, N, K, m.group(1), m.group(2), S, times["production"] * 1e3, bu.hgemm_flops(M, N, K) / times["production"] * 1e-9, times["two launches"] * 1e3, times["one launch"] * 1e3,
           100.0 * (times["two launches"] / times["one launch"] - 1.0), err, "ok" if ok else "WRONG", same, "one launch" if "in-kernel fix-up" in plan else "two launches"), flush=True)
    del a, b, outs
