"""Launch loop for `rocprofv3 --kernel-trace --stats`: every bandwidth-bound kernel family at a reference shape,
25 launches each, so the stats CSV carries one row per kernel with its average device time. bw_prof_summary.py
turns that into GB/s with the algorithmic bytes of SURVEY 8(d)."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import _loader  # noqa: E402

dev = torch.device("cuda:0")
so = _loader.load_so("libcln_amd.so")
st = lambda: torch.cuda.current_stream().cuda_stream
S = K = 4096
x = torch.randn(S, K, device=dev)
y = torch.zeros_like(x)
xh, yh = x.half(), y.half()
x2 = torch.randn(S, K, device=dev)
x2h = x2.half()
z = torch.zeros(1, device=dev)
idx = torch.randint(0, 4096, (4096,), device=dev, dtype=torch.int32)
w = torch.randn(4096, 1024, device=dev)
o = torch.zeros(4096, 1024, device=dev)
hist_in = torch.randint(0, 1024, (1 << 24,), device=dev, dtype=torch.int32)
hist_out = torch.zeros(1024, device=dev, dtype=torch.int32)
f = ctypes.c_float
n = x.numel()
CALLS = [  # (tag, substring of the kernel name, algorithmic bytes, callable)
    ("elementwise_add_f32x4", "add", 3 * n * 4, lambda: so.elementwise_add_f32x4(x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, st())),
    ("elementwise_add_f16x8_pack", "add", 3 * n * 2, lambda: so.elementwise_add_f16x8_pack(xh.data_ptr(), x2h.data_ptr(), yh.data_ptr(), n, st())),
    ("block_all_reduce_sum_f32x4_f32", "reduce_sum", n * 4, lambda: so.block_all_reduce_sum_f32x4_f32(x.data_ptr(), z.data_ptr(), n, st())),
    ("block_all_reduce_sum_f16x8_pack_f32", "reduce_sum", n * 2, lambda: so.block_all_reduce_sum_f16x8_pack_f32(xh.data_ptr(), z.data_ptr(), n, st())),
    ("safe_softmax_f32x4_per_token", "softmax", 2 * n * 4, lambda: so.safe_softmax_f32x4_per_token(x.data_ptr(), y.data_ptr(), S, K, st())),
    ("safe_softmax_f16x8_pack_f32_per_token", "softmax", 2 * n * 2, lambda: so.safe_softmax_f16x8_pack_f32_per_token(xh.data_ptr(), yh.data_ptr(), S, K, st())),
    ("layer_norm_f32x4", "layer_norm", 2 * n * 4, lambda: so.layer_norm_f32x4(x.data_ptr(), y.data_ptr(), f(1.0), f(0.0), S, K, st())),
    ("layer_norm_f16x8_pack_f32", "layer_norm", 2 * n * 2, lambda: so.layer_norm_f16x8_pack_f32(xh.data_ptr(), yh.data_ptr(), f(1.0), f(0.0), S, K, st())),
    ("rms_norm_f32x4", "rms_norm", 2 * n * 4, lambda: so.rms_norm_f32x4(x.data_ptr(), y.data_ptr(), f(1.0), S, K, st())),
    ("rms_norm_f16x8_pack_f32", "rms_norm", 2 * n * 2, lambda: so.rms_norm_f16x8_pack_f32(xh.data_ptr(), yh.data_ptr(), f(1.0), S, K, st())),
    ("rope_f32x4_pack", "rope", 2 * n * 4, lambda: so.rope_f32x4_pack(x.data_ptr(), y.data_ptr(), S, K, 0, st())),
    ("gelu_f32x4", "unary", 2 * n * 4, lambda: so.gelu_f32x4(x.data_ptr(), y.data_ptr(), n, st())),
    ("relu_f16x8_pack", "unary", 2 * n * 2, lambda: so.relu_f16x8_pack(xh.data_ptr(), yh.data_ptr(), n, st())),
    ("mat_transpose_f32x4_shared_bcf_col2row2d", "tr_lds", 2 * n * 4, lambda: so.mat_transpose_f32x4_shared_bcf_col2row2d(x.data_ptr(), y.data_ptr(), S, K, st())),
    ("embedding_f32x4_pack", "embedding", 2 * o.numel() * 4, lambda: so.embedding_f32x4_pack(idx.data_ptr(), w.data_ptr(), o.data_ptr(), 4096, 1024, 4096, st())),
    ("histogram_i32x4", "histogram", hist_in.numel() * 4, lambda: so.histogram_i32x4(hist_in.data_ptr(), hist_out.data_ptr(), hist_in.numel(), 1024, st())),
]
# steady-state rows: the scripts' own large shape [8192, 8192] -- a launch is 4x longer, the ~2 us ramp and
# (reduce) the ~3 us fan-in of 256 device-scope atomics weigh a quarter as much; fp16 = 134 MB, fp32 = 268 MB (past the 256 MiB
# Infinity Cache). Allocated after the 4096^2 rows have run.
S2 = K2 = 8192


def big_calls():
    xb = torch.randn(S2, K2, device=dev)
    yb = torch.zeros_like(xb)
    xbh, ybh = xb.half(), yb.half()
    x2bh = torch.randn(S2, K2, device=dev, dtype=torch.half)
    nb = xb.numel()
    keep = (xb, yb, xbh, ybh, x2bh)
    return keep, [
        ("elementwise_add_f16x8_pack@8192", "add", 3 * nb * 2, lambda: so.elementwise_add_f16x8_pack(xbh.data_ptr(), x2bh.data_ptr(), ybh.data_ptr(), nb, st())),
        ("block_all_reduce_sum_f32x4_f32@8192", "reduce_sum", nb * 4, lambda: so.block_all_reduce_sum_f32x4_f32(xb.data_ptr(), z.data_ptr(), nb, st())),
        ("block_all_reduce_sum_f16x8_pack_f32@8192", "reduce_sum", nb * 2, lambda: so.block_all_reduce_sum_f16x8_pack_f32(xbh.data_ptr(), z.data_ptr(), nb, st())),
        ("safe_softmax_f32x4_per_token@8192", "softmax", 2 * nb * 4, lambda: so.safe_softmax_f32x4_per_token(xb.data_ptr(), yb.data_ptr(), S2, K2, st())),
        ("safe_softmax_f16x8_pack_f32_per_token@8192", "softmax", 2 * nb * 2, lambda: so.safe_softmax_f16x8_pack_f32_per_token(xbh.data_ptr(), ybh.data_ptr(), S2, K2, st())),
        ("layer_norm_f16x8_pack_f32@8192", "layer_norm", 2 * nb * 2, lambda: so.layer_norm_f16x8_pack_f32(xbh.data_ptr(), ybh.data_ptr(), f(1.0), f(0.0), S2, K2, st())),
        ("rms_norm_f16x8_pack_f32@8192", "rms_norm", 2 * nb * 2, lambda: so.rms_norm_f16x8_pack_f32(xbh.data_ptr(), ybh.data_ptr(), f(1.0), S2, K2, st())),
        ("relu_f16x8_pack@8192", "unary", 2 * nb * 2, lambda: so.relu_f16x8_pack(xbh.data_ptr(), ybh.data_ptr(), nb, st())),
        ("rope_f32x4_pack@8192", "rope", 2 * nb * 4, lambda: so.rope_f32x4_pack(xb.data_ptr(), yb.data_ptr(), S2, K2, 0, st())),
    ]


order = []


def run(calls):
    for tag, sub, nbytes, fn in calls:
        for _ in range(25):
            rc = fn()
            assert rc == 0, (tag, rc)
        torch.cuda.synchronize()
        order.append({"tag": tag, "kernel_substring": sub, "bytes": nbytes, "launches": 25})


run(CALLS)
# larger gathers / counts for the indexing rows: 64 Ki rows of 1024 floats (268 MB out), 2^26 histogram inputs (268 MB)
idx2 = torch.randint(0, 65536, (65536,), device=dev, dtype=torch.int32)
w2 = torch.randn(65536, 1024, device=dev)
o2 = torch.zeros(65536, 1024, device=dev)
hist_in2 = torch.randint(0, 1024, (1 << 26,), device=dev, dtype=torch.int32)
run([("embedding_f32x4_pack@65536x1024", "embedding", 2 * o2.numel() * 4, lambda: so.embedding_f32x4_pack(idx2.data_ptr(), w2.data_ptr(), o2.data_ptr(), 65536, 1024, 65536, st())),
     ("histogram_i32x4@2^26", "histogram", hist_in2.numel() * 4, lambda: so.histogram_i32x4(hist_in2.data_ptr(), hist_out.data_ptr(), hist_in2.numel(), 1024, st()))])
del w2, o2, hist_in2, w, o, hist_in
keep, calls = big_calls()
run(calls)
# round 4: the SAME rows bench.py prints as `gbps` (configs.bandwidth): every kernel over ROTATING buffer sets whose combined footprint is four times the
# Infinity Cache, so the traced kernel durations are HBM durations (the rows above re-use one buffer set: bench.py's `gbps_same_buffers`)
del keep, calls
torch.cuda.empty_cache()
import bench_configs as bc  # noqa: E402  (repository root; only its kernel table and launch closures are used here, no CPU leg)
zz = torch.zeros(4, dtype=torch.float32, device=dev)
for name, dtype, kind, bpe, _cpu in bc._bw_specs():
    fn = _loader.symbol(name)
    for (Sr, Kr) in ((4096, 4096), (8192, 8192)):
        nr = Sr * Kr
        esz = 2 if dtype == torch.float16 else 4
        n_in, n_out = (2 if kind == "P3" else 1), (0 if kind == "R1" else 1)
        set_bytes = (n_in + n_out) * nr * esz
        nsets = max(3, (bc.ROTATE_FOOTPRINT + set_bytes - 1) // set_bytes)
        pool = torch.randn(nsets * (n_in + n_out), Sr, Kr, device=dev, dtype=dtype)
        rot = [bc._bw_call(fn, kind, pool[i * (n_in + n_out)], pool[i * (n_in + n_out) + 1] if n_in == 2 else None,
                           pool[i * (n_in + n_out) + n_in] if n_out else None, zz, Sr, Kr) for i in range(nsets)]
        launches = 0
        for _ in range(3):
            for c in rot:
                assert c() == 0, name
                launches += 1
        torch.cuda.synchronize()
        order.append({"tag": "%s@%d rotating x%d" % (name, Sr, nsets), "kernel_substring": "", "bytes": bpe * nr, "launches": launches})
        del pool, rot
        torch.cuda.empty_cache()
out = os.environ.get("BW_PROF_ORDER", os.path.join(ROOT, "gpurun_out", "bw_prof_order.json"))
json.dump(order, open(out, "w"), indent=1)
