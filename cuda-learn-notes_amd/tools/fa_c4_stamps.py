"""Round-6 probe (VERDICT r5 #1b): where the FIXED cost of the attention launch at config C4 goes. Runs the shipped D = 64 / D = 128 kernels in their
M16X_STAMP form (csrc/flash_attn_m16x.cuh: every wave stamps s_memrealtime -- 100 MHz, chip-wide -- and s_memtime at entry, top of the KV loop, end of
the loop, last O store landed; probe codes 988 / 989 = 800 + 188 / 189) and prints, per launch:
  * the dispatch ramp: first wave's entry -> last wave's entry, per XCD
  * prologue (entry -> KV loop), loop, epilogue per wave: min / median / max
  * the launch's span (first entry -> last exit), the sum of the medians, and the finish-time spread (what a back-filling scheme could recover)
  * per-tile loop time from two sequence lengths (N and 2N: the difference is pure loop) -> fixed cost = span - T * per-tile
Lines start with STAMP.   python fa_c4_stamps.py [B H N D]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import _loader, host, bench_utils as bu  # noqa: E402

dev = torch.device("cuda:0")
probe = _loader.load_so("libcln_amd_probe.so")
probe.cln_probe_set_stamps.argtypes, probe.cln_probe_set_stamps.restype = [ctypes.c_void_p], ctypes.c_int
fa = pkg.flash_attn_lib()


def med(t):
    return t.float().median().item()


def run(B, H, N, D, code, rows_per_wg, reps=5):
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    ref = torch.zeros_like(q)
    fa.flash_attn_mma_stages_split_q_shared_qkv(q, k, v, ref, 2)
    nwg = B * H * N // rows_per_wg
    nw = nwg * 8
    buf = torch.zeros(nw * 10, dtype=torch.int64, device=dev)
    assert probe.cln_probe_set_stamps(buf.data_ptr()) == 0
    out = []
    for r in range(reps + 2):
        buf.zero_()
        torch.cuda.synchronize()
        host.fa2_variant((8, 0, 0, code), q, k, v, o)
        torch.cuda.synchronize()
        if r >= 2:
            out.append(buf.view(nw, 10).cpu().clone())
    probe.cln_probe_set_stamps(None)
    same = torch.equal(o, ref)
    # plain timing of the unstamped kernel for reference
    ms = bu.time_region_events(lambda: fa.flash_attn_mma_stages_split_q_shared_qkv(q, k, v, o, 2), 200)
    print("STAMP shape [%d,%d,%d,%d] workgroups %d (%.2f per CU)  stamped form bit-identical to production: %s  production %.2f us/launch (event region of 200)"
          % (B, H, N, D, nwg, nwg / 256.0, same, ms * 1e3), flush=True)
    spans = []
    for s in out:
        rt = s[:, 0:8:2].double() * 10.0  # ns (100 MHz)
        mt = s[:, 1:8:2].double()
        t0 = rt[:, 0].min()
        entry_, loop0, loop1, exit_ = (rt[:, i] - t0 for i in range(4))
        span = exit_.max().item()
        spans.append(span)
        clk = ((mt[:, 2] - mt[:, 1]) / (rt[:, 2] - rt[:, 1]).clamp(min=1)).median().item()  # shader ticks per ns inside the loop
    s = out[-1]
    rt = s[:, 0:8:2].double() * 10.0
    mt = s[:, 1:8:2].double()
    t0 = rt[:, 0].min()
    entry_, loop0, loop1, exit_ = (rt[:, i] - t0 for i in range(4))
    xcc = (s[:, 9] & 0xF)
    T = N // (128 if rows_per_wg == 256 else 64)
    print("STAMP   launch span (first wave in -> last wave out) over %d launches: %s ns" % (len(spans), " ".join("%.0f" % x for x in spans)))
    print("STAMP   wave entry (dispatch ramp): median %.0f  max %.0f ns;  per XCD max: %s" % (med(entry_), entry_.max().item(),
          " ".join("%d:%.0f" % (x, entry_[xcc == x].max().item()) for x in sorted(set(xcc.tolist())))))
    pro, loop, epi = loop0 - entry_, loop1 - loop0, exit_ - loop1
    for name, t in (("prologue (entry -> loop)", pro), ("KV loop (%d tiles)" % T, loop), ("epilogue (loop end -> O landed)", epi), ("wave lifetime", exit_ - entry_)):
        print("STAMP   %-34s min %7.0f  median %7.0f  max %7.0f ns" % (name, t.min().item(), med(t), t.max().item()))
    print("STAMP   per tile (median loop / %d): %.1f ns;  shader clock inside the loop: %.3f GHz (s_memtime ticks per s_memrealtime ns: %.3f)" % (T, med(loop) / T, clk * 1.0, clk))
    print("STAMP   finish: first wave out %.0f, median %.0f, last %.0f ns  -> spread %.0f ns (%.1f %% of the span)" % (exit_.min().item(), med(exit_), exit_.max().item(),
          exit_.max().item() - exit_.min().item(), 100.0 * (exit_.max().item() - exit_.min().item()) / spans[-1]))
    # per XCD: is the finish spread systematic (a slower die) or random?
    for xi in sorted(set(xcc.tolist())):
        sel = xcc == xi
        print("STAMP   XCD %d: %4d waves  entry median %5.0f  loop median %7.0f (per tile %6.1f)  exit median %7.0f  max %7.0f ns   clock %.3f GHz"
              % (xi, int(sel.sum()), med(entry_[sel]), med(loop[sel]), med(loop[sel]) / T, med(exit_[sel]), exit_[sel].max().item(),
                 ((mt[sel, 2] - mt[sel, 1]) / (rt[sel, 2] - rt[sel, 1]).clamp(min=1)).median().item()))
    # per workgroup: the slowest and the fastest
    wg_exit = exit_.view(-1, 8).max(dim=1).values
    wg_loop = loop.view(-1, 8).float().mean(dim=1)
    order = wg_exit.argsort()
    print("STAMP   workgroup exit: fastest 5 %s  slowest 5 %s ns; loop time of those: %s / %s" % (
        [int(wg_exit[i]) for i in order[:5]], [int(wg_exit[i]) for i in order[-5:]], [int(wg_loop[i]) for i in order[:5]], [int(wg_loop[i]) for i in order[-5:]]))
    # group 0 / group 1 of a workgroup (waves 0-3 / 4-7): group 1 runs one phase behind
    g = torch.arange(nw) % 8 // 4
    for gi in (0, 1):
        print("STAMP   group %d: prologue median %.0f, loop median %.0f, epilogue median %.0f, exit median %.0f ns" % (gi, med(pro[g == gi]), med(loop[g == gi]), med(epi[g == gi]), med(exit_[g == gi])))
    return spans[-1], med(loop), T, ms * 1e3


if __name__ == "__main__":
    args = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else None
    shapes = [tuple(args)] if args else [(4, 8, 2048, 64), (4, 8, 2048, 128), (4, 8, 4096, 128)]
    res = {}
    for (B, H, N, D) in shapes:
        res[(B, H, N, D)] = run(B, H, N, D, 988, 256)
    for D in (64, 128):
        a, b = res.get((4, 8, 2048, D)), res.get((4, 8, 4096, D))
        if a and not b:
            print("STAMP D=%d: C4-shaped launch span %.0f ns = %d tiles x %.1f ns + %.0f ns outside the KV loop" % (D, a[0], a[2], a[1] / a[2], a[0] - a[1]))
        if a and b:
            # same workgroup count per unit of N? no: 2N doubles the workgroups (2 rounds) AND the tiles; per-tile from the loop medians instead
            per_tile = b[1] / b[2]
            print("STAMP D=%d: per-tile loop time %.1f ns (N=4096: 2 rounds of workgroups share the CUs) vs %.1f ns at N=2048 (1 round); C4 span %.0f ns = %d tiles x %.1f + %.0f ns fixed"
                  % (D, per_tile, a[1] / a[2], a[0], a[2], a[1] / a[2], a[0] - a[1]))
