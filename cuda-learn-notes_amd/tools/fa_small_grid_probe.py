"""Small grids at D = 64 / 128 (fewer 256-row workgroups than CUs): the v2 kernel with 2 / 4 / 8 waves per workgroup against the
two-group ping-pong kernel fa2_fwd_m16x (8 waves x 32 rows), and what the planner picks.  Output feeds the thresholds in
csrc/flash_attn.hip fa2_plan.  Run on the GPU box: python cuda-learn-notes_amd/tools/fa_small_grid_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host  # noqa: E402

fa = pkg.flash_attn_lib()
dev = torch.device("cuda:0")
fn = fa.flash_attn_mma_stages_split_q_shared_qkv
V2_OPT = {32: 16397, 64: 16397, 96: 16399, 128: 16399, 256: 15}


def tf(call, flops, n=120):
    bu.prewarm(call, 0.08)
    ms = bu.time_region_events(call, n)
    return flops / ms * 1e-9


for D in (32, 64, 96, 128, 256):
    for N in (1024, 2048, 4096):
        for wgs in (32, 64, 96, 128, 160, 192, 224, 256):
            bh = wgs * 256 // N
            if bh * N != wgs * 256:
                continue
            shape = (1, bh, N, D)
            q, k, v = (torch.randn(*shape, dtype=torch.half, device=dev) for _ in range(3))
            o = torch.zeros_like(q)
            fl = bu.mha_flops_conventional(*shape)
            row = {}
            for nw in (2, 4, 8):
                try:
                    row["v2x%d" % nw] = tf(lambda: host.fa2_variant((nw, 0, V2_OPT[D], 0), q, k, v, o), fl)
                except RuntimeError:
                    row["v2x%d" % nw] = float("nan")
            try:  # the two-group kernels: fa2_fwd_m16x at D = 64 / 128 (probe code 853), the 16x16x32 pair kernel at D = 256 (544)
                row["m16x"] = tf(lambda: host.fa2_variant((8, 0, 0, 853 if D <= 128 else 544), q, k, v, o), fl)
            except RuntimeError:
                row["m16x"] = float("nan")
            row["plan"] = tf(lambda: fn(q, k, v, o, 2), fl)
            best = max((k2 for k2 in row if k2 != "plan"), key=lambda k2: row[k2] if row[k2] == row[k2] else -1)
            print("SMALLGRID D=%3d N=%4d BH=%3d wgs256=%3d  v2x2 %6.1f  v2x4 %6.1f  v2x8 %6.1f  m16x %6.1f | plan %6.1f (%s)  best %s %+.1f%%"
                  % (D, N, bh, wgs, row["v2x2"], row["v2x4"], row["v2x8"], row["m16x"], row["plan"],
                     pkg.manifest.describe(fn.__name__, shape, 2).split("<")[0] + " " + pkg.manifest.describe(fn.__name__, shape, 2).split("> ")[1][:7],
                     best, (row[best] / row["plan"] - 1) * 100), flush=True)
