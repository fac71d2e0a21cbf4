"""Host-side helpers of the bench drivers, re-authored from the reference's behaviour:
  make_block_swizzle_stride  kernels/hgemm/hgemm.py:71-81      (N-band policy)
  as_col_major               kernels/hgemm/tools/utils.py:135-140 (TN operand maker)
  get_device_name / pretty_print_line  kernels/hgemm/tools/utils.py:7-15, :96-101
  get_mha_tflops             kernels/flash-attn/flash_attn_mma.py:191-222 (FLOP model)
  timing protocol            kernels/hgemm/hgemm.py:115-138 (warmup, synchronize, time.time(), iters)
plus MI355X roofline constants (/opt/skills/guides/MI355X_MICROARCH.md chip table).
"""
import time

import torch

PEAK_FP16_MFMA_TFLOPS = 2500.0  # dense fp16/bf16 MFMA, MI355X
PEAK_HBM_GBPS = 8000.0          # HBM3E spec (6290 measured float4 copy)


def get_device_name():
    if not torch.cuda.is_available():
        return "cpu"
    return torch.cuda.get_device_name(torch.cuda.current_device()).replace(" ", "_")


def pretty_print_line(m: str = "", sep: str = "-", width: int = 150):
    res_len = width - len(m)
    left = res_len // 2
    print(sep * left + m + sep * (res_len - left))


def make_block_swizzle_stride(N: int, K: int, swizzle_factor: float = None) -> int:
    if swizzle_factor is None:
        swizzle_factor = 0.5 if N <= 4096 else 0.25
        if all((N >= 14848, K > 8192, N % 8 == 0)):
            swizzle_factor = 0.125
    swizzle_stride = int(N * swizzle_factor)
    return swizzle_stride if swizzle_stride >= 256 else 1


@torch.no_grad()
def as_col_major(x: torch.Tensor) -> torch.Tensor:
    return x.t().reshape(x.shape).contiguous()


def hgemm_flops(M, N, K):
    return 2.0 * M * N * K


def hgemm_bytes(M, N, K):
    return 2.0 * (M * K + K * N + M * N)


def get_mha_tflops(B, H, N, D, secs=1.0, only_matmul=False):
    flops_qk = B * H * N * N * (2 * D - 1)
    flops_scaling = B * H * N * N
    flops_row_max = B * H * N * (N - 1)
    flops_subtract_max = B * H * N * N
    flops_exp = B * H * N * N
    flops_row_sum = B * H * N * (N - 1)
    flops_normalization = B * H * N * N
    flops_safe_softmax = flops_row_max + flops_subtract_max + flops_exp + flops_row_sum + flops_normalization
    flops_pv = B * H * N * D * (2 * N - 1)
    total = flops_qk + flops_scaling + flops_safe_softmax + flops_pv
    if only_matmul:
        total = flops_qk + flops_pv
    return total * 1e-12 / secs


def mha_flops_conventional(B, H, N, D):
    return 4.0 * B * H * N * N * D


def time_call(fn, warmup: int, iters: int) -> float:
    """Mean seconds per call with the reference's protocol (host clock around `iters` async launches)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters


def time_call_events(fn, warmup: int, iters: int, stream=None):
    """Per-launch device time (ms) from HIP events recorded on the stream the kernels run on."""
    stream = stream or torch.cuda.current_stream()
    for _ in range(warmup):
        fn()
    start = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    end = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        start[i].record(stream)
        fn()
        end[i].record(stream)
    torch.cuda.synchronize()
    ts = [s.elapsed_time(e) for s, e in zip(start, end)]
    return sum(ts) / len(ts), min(ts), ts
