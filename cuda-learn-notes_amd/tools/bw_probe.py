"""GPU probe of the bandwidth kernels with hipGraph timing (kernel time without the Python launch path).
Prints GB/s against the algorithmic bytes of SURVEY 8(d) next to the stock torch op."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host, _loader  # noqa: E402

dev = torch.device("cuda:0")
so = _loader.load_so("libcln_amd.so")
st = lambda: torch.cuda.current_stream().cuda_stream


def row(tag, fn, nbytes):
    try:
        ms, best = bu.time_call_graph(fn, 20, 5)
        print("BW %-44s %8.2f us  %7.1f GB/s (best %7.1f)" % (tag, ms * 1e3, nbytes / ms * 1e-6, nbytes / best * 1e-6), flush=True)
    except Exception as e:
        print("BW", tag, "ERR", str(e)[:150], flush=True)


def raw(name, *args):
    fn = getattr(so, name)
    return lambda: fn(*args, st())


ew = pkg.load("elementwise")
for S, K in ((2048, 2048), (4096, 4096)):
    a = torch.randn(S, K, device=dev); b = torch.randn(S, K, device=dev); c = torch.zeros_like(a)
    nb = 3 * a.numel() * 4
    for n in ("elementwise_add_f32", "elementwise_add_f32x4"):
        row("%s [%d,%d]" % (n, S, K), raw(n, a.data_ptr(), b.data_ptr(), c.data_ptr(), a.numel()), nb)
    row("torch.add f32 [%d,%d]" % (S, K), lambda: torch.add(a, b, out=c), nb)
    ah, bh, ch = a.half(), b.half(), c.half()
    for n in ("elementwise_add_f16", "elementwise_add_f16x2", "elementwise_add_f16x8", "elementwise_add_f16x8_pack"):
        row("%s [%d,%d]" % (n, S, K), raw(n, ah.data_ptr(), bh.data_ptr(), ch.data_ptr(), ah.numel()), nb // 2)
    row("torch.add f16 [%d,%d]" % (S, K), lambda: torch.add(ah, bh, out=ch), nb // 2)

for S, K in ((2048, 2048), (4096, 4096)):
    x = torch.randn(S, K, device=dev)
    y = torch.zeros(1, device=dev)
    yi = torch.zeros(1, device=dev, dtype=torch.int32)
    for n, t in (("block_all_reduce_sum_f32_f32", x), ("block_all_reduce_sum_f32x4_f32", x),
                 ("block_all_reduce_sum_f16_f32", x.half()), ("block_all_reduce_sum_f16x8_pack_f32", x.half()),
                 ("block_all_reduce_sum_f16x8_pack_f16", x.half()), ("block_all_reduce_sum_bf16x8_pack_f32", x.bfloat16()),
                 ("block_all_reduce_sum_fp8_e4m3x16_pack_f16", x.to(torch.float8_e4m3fn)),
                 ("block_all_reduce_sum_i8_i32", x.to(torch.int8)), ("block_all_reduce_sum_i8x16_pack_i32", x.to(torch.int8))):
        out = yi if "i32" in n else y
        row("%s [%d,%d]" % (n, S, K), raw(n, t.data_ptr(), out.data_ptr(), t.numel()), t.numel() * t.element_size())
    row("torch.sum f32 [%d,%d]" % (S, K), lambda: torch.sum(x), x.numel() * 4)
    xh = x.half()
    row("torch.sum f16 [%d,%d]" % (S, K), lambda: torch.sum(xh), x.numel() * 2)

for H in (256, 1024, 4096, 8192):
    S = 4096
    x = torch.randn(S, H, device=dev); y = torch.zeros_like(x)
    nb = 2 * x.numel() * 4
    names = ["safe_softmax_f32x4_per_token", "online_safe_softmax_f32x4_pack_per_token", "softmax_f32x4_per_token"]
    if H <= 1024:
        names = ["softmax_f32_per_token", "safe_softmax_f32_per_token"] + names
    for n in names:
        row("%s [%d,%d]" % (n, S, H), raw(n, x.data_ptr(), y.data_ptr(), S, H), nb)
    row("torch.softmax f32 [%d,%d]" % (S, H), lambda: torch.softmax(x, dim=1, out=y), nb)
    xh, yh = x.half(), y.half()
    row("safe_softmax_f16x8_pack_f32_per_token [%d,%d]" % (S, H), raw("safe_softmax_f16x8_pack_f32_per_token", xh.data_ptr(), yh.data_ptr(), S, H), nb // 2)
    row("torch.softmax f16 [%d,%d]" % (S, H), lambda: torch.softmax(xh, dim=1, out=yh), nb // 2)

import ctypes
for N_, K in ((4096, 512), (4096, 1024), (4096, 4096), (8192, 8192)):
    x = torch.randn(N_, K, device=dev); y = torch.zeros_like(x)
    nb = 2 * x.numel() * 4
    if K <= 4096:
        row("layer_norm_f32x4 [%d,%d]" % (N_, K), raw("layer_norm_f32x4", x.data_ptr(), y.data_ptr(), ctypes.c_float(1.0), ctypes.c_float(0.0), N_, K), nb)
        row("rms_norm_f32x4 [%d,%d]" % (N_, K), raw("rms_norm_f32x4", x.data_ptr(), y.data_ptr(), ctypes.c_float(1.0), N_, K), nb)
    xh, yh = x.half(), y.half()
    for n in ("layer_norm_f16x8_pack_f16", "layer_norm_f16x8_pack_f32"):
        row("%s [%d,%d]" % (n, N_, K), raw(n, xh.data_ptr(), yh.data_ptr(), ctypes.c_float(1.0), ctypes.c_float(0.0), N_, K), nb // 2)
    for n in ("rms_norm_f16x8_pack_f16", "rms_norm_f16x8_pack_f32"):
        row("%s [%d,%d]" % (n, N_, K), raw(n, xh.data_ptr(), yh.data_ptr(), ctypes.c_float(1.0), N_, K), nb // 2)
    row("torch layer_norm f16 [%d,%d]" % (N_, K), lambda: torch.nn.functional.layer_norm(xh, (K,)), nb // 2)

for M_, N_ in ((4096, 512), (8192, 1024)):
    x = torch.randn(M_, N_, device=dev); y = torch.zeros_like(x)
    nb = 2 * x.numel() * 4
    for n in ("rope_f32", "rope_f32x4_pack"):
        row("%s [%d,%d]" % (n, M_, N_), raw(n, x.data_ptr(), y.data_ptr(), M_, N_, 0), nb)
