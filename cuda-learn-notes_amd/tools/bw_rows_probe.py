"""GPU probe: the row kernels (softmax / layer-norm / rms-norm) over the reference scripts' whole shape set (S = 4096, row length 256 ... 8192;
softmax.py:150-230, layer_norm.py:66-185, rms_norm.py) -- the widest rung of each family against torch's own GPU kernel for the same op
(torch.softmax / torch.nn.functional.layer_norm / rms via torch ops), launch-inclusive event-timed regions over the same buffers, and the
algorithmic GB/s.  Finds row lengths where the launch shape is off.  python bw_rows_probe.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu  # noqa: E402

dev = torch.device("cuda:0")
sm, ln, rn = pkg.load("softmax"), pkg.load("layer_norm"), pkg.load("rms_norm")
S = 4096


def us(fn, n=300):
    bu.prewarm(fn, 0.03)
    return min(bu.time_region_events(fn, n), bu.time_region_events(fn, n)) * 1e3


for H in (256, 512, 1024, 2048, 4096, 8192):
    x = torch.randn(S, H, device=dev)
    o = torch.zeros_like(x)
    xh, oh = x.half(), torch.zeros(S, H, dtype=torch.half, device=dev)
    w32, w16 = torch.ones(H, device=dev), torch.ones(H, dtype=torch.half, device=dev)
    rows = [
        ("softmax f32x4(safe)", lambda: sm.safe_softmax_f32x4_per_token(x, o), lambda: torch.softmax(x, dim=1, out=o), 8 * S * H),
        ("softmax f32x4(per)", lambda: sm.softmax_f32x4_per_token(x, o), lambda: torch.softmax(x, dim=1, out=o), 8 * S * H),
        ("softmax f32x4(online)", lambda: sm.online_safe_softmax_f32x4_pack_per_token(x, o), lambda: torch.softmax(x, dim=1, out=o), 8 * S * H),
        ("softmax f16x8pack(safe)", lambda: sm.safe_softmax_f16x8_pack_f32_per_token(xh, oh), lambda: torch.softmax(xh, dim=1, out=oh), 4 * S * H),
        ("layer_norm f32x4", lambda: ln.layer_norm_f32x4(x, o, 1.0, 0.0), lambda: F.layer_norm(x, (H,)), 8 * S * H),
        ("layer_norm f16x8pack_f16", lambda: ln.layer_norm_f16x8_pack_f16(xh, oh, 1.0, 0.0), lambda: F.layer_norm(xh, (H,)), 4 * S * H),
        ("layer_norm f16x8pack_f32", lambda: ln.layer_norm_f16x8_pack_f32(xh, oh, 1.0, 0.0), lambda: F.layer_norm(xh, (H,)), 4 * S * H),
        ("rms_norm f32x4", lambda: rn.rms_norm_f32x4(x, o, 1.0), lambda: F.rms_norm(x, (H,)), 8 * S * H),
        ("rms_norm f16x8pack_f16", lambda: rn.rms_norm_f16x8_pack_f16(xh, oh, 1.0), lambda: F.rms_norm(xh, (H,)), 4 * S * H),
        ("rms_norm f16x8pack_f32", lambda: rn.rms_norm_f16x8_pack_f32(xh, oh, 1.0), lambda: F.rms_norm(xh, (H,)), 4 * S * H),
    ]
    for tag, ours, th, nbytes in rows:
        try:
            t_o = us(ours)
        except (RuntimeError, AttributeError) as e:
            print("BWROWS H=%5d %-26s n/a %s" % (H, tag, str(e)[:70]), flush=True)
            continue
        t_t = us(th)
        print("BWROWS H=%5d %-26s ours %7.2f us %7.1f GB/s | torch %7.2f us | ours/torch %.2f" % (H, tag, t_o, nbytes / t_o * 1e-3, t_t, t_o / t_t), flush=True)
