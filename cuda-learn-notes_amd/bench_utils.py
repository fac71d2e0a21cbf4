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
# what `roofline.traffic` counts: rocprofv3 FETCH_SIZE / WRITE_SIZE derive from the L2's fabric-side request counters
# (TCC_EA0_RDREQ / WRREQ): bytes between the XCD L2s and the memory fabric. Infinity-Cache hits are INSIDE the number, so
# it is an upper bound of the HBM bytes, not the HBM bytes (MI355X_MICROARCH.md, HBM / Infinity Cache sections).
TRAFFIC_IS = "L2<->fabric bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); Infinity-Cache hits included: upper bound of HBM bytes"
HEADLINE_HGEMM_NAME = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"  # what it runs: manifest.describe(name, dims, stages)


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


def time_call_graph(fn, iters: int = 20, replays: int = 5):
    """Per-launch device time (ms) for kernels shorter than the host launch path: capture `iters` calls in a
    hipGraph on a side stream, replay it, divide. Includes the ~1.5 us dependent-kernel boundary."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best, tot = 1e30, 0.0
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters
        best, tot = min(best, t), tot + t
    return tot / replays, best


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """bench.py timing contract: the step time of the job is the MAX over ranks (replicas run independently;
    no data-path collective). `dist` is torch.distributed or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_value(per_rank_units: float, world: int, seconds: float) -> float:
    """Whole-job throughput of N independent replicas: units all ranks processed / max-over-ranks time."""
    return world * per_rank_units / seconds


def prewarm(fn, seconds: float):
    """Back-to-back launches for `seconds` of wall time (untimed): brings the chip to its sustained power/clock state."""
    if seconds <= 0:
        return
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()


def settle(fn, batch: int, tol: float = 0.03, max_seconds: float = 3.0, min_seconds: float = 0.2):
    """Untimed steady-state check after the time-based pre-warm: event-timed batches of `batch` back-to-back launches
    until the last THREE batches agree within `tol`, the last one is within `tol` of the fastest seen and at least
    `min_seconds` have been observed (or `max_seconds` pass). On some boxes the launch bursts of a process run at ~2/3 of
    the sustained rate for a while (a 20-step region read 955 instead of 1430-1460 TF once in round 2, directly after the
    full GPU suite). Returns the ms-per-launch history; a plateau that outlasts `max_seconds` is reported as it is."""
    hist = []
    t0 = time.time()
    while time.time() - t0 < max_seconds:
        hist.append(time_region_events(fn, batch))
        if len(hist) >= 3 and time.time() - t0 >= min_seconds:
            last = hist[-3:]
            if max(last) <= (1.0 + tol) * min(last) and hist[-1] <= (1.0 + tol) * min(hist):
                break
    return hist


def time_region_events(fn, iters: int, stream=None) -> float:
    """Mean ms per launch from ONE pair of HIP events around `iters` back-to-back launches on the launch stream
    (per-launch event pairs insert an idle gap after every kernel and read 5-15 % low on short kernels)."""
    stream = stream or torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def sdpa_rows(q, k, v, side_ms):
    """torch SDPA with each backend forced in turn: {backend: TFLOPS (4BHN^2D) | error text}, plus the default
    dispatch. Names the implementation behind the comparison row."""
    import torch.nn.functional as F
    B, H, N, D = q.shape
    fl = mha_flops_conventional(B, H, N, D)
    rows = {}
    iters = 30 if N <= 2048 else 8
    try:
        ms = side_ms(lambda: F.scaled_dot_product_attention(q, k, v), iters)
        rows["default_dispatch"] = round(fl / (ms * 1e-3) * 1e-12, 2)
    except Exception as e:
        rows["default_dispatch"] = "error: " + str(e)[:80]
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel
        for name, be in (("flash_attention", SDPBackend.FLASH_ATTENTION), ("efficient_attention", SDPBackend.EFFICIENT_ATTENTION)):
            try:
                with sdpa_kernel(be):
                    ms = side_ms(lambda: F.scaled_dot_product_attention(q, k, v), iters)
                rows[name] = round(fl / (ms * 1e-3) * 1e-12, 2)
            except Exception as e:
                rows[name] = "unavailable: " + str(e).split("\n")[0][:80]
    except Exception as e:
        rows["backends"] = "torch.nn.attention unavailable: " + str(e)[:80]
    rows["enabled_flags"] = {"flash": bool(torch.backends.cuda.flash_sdp_enabled()),
                             "mem_efficient": bool(torch.backends.cuda.mem_efficient_sdp_enabled()),
                             "math": bool(torch.backends.cuda.math_sdp_enabled())}
    return rows


def pmc_value(profiles_dir: str, stem: str, key: str, kernel_sub: str = ""):
    """`key` of the newest committed rocprofv3 PMC summary profiles/rNN_<stem>.json (written by tools/pmc_summary.py on
    the GPU box: PMC counters cannot be collected from inside the timed process). `kernel_sub`: only entries whose
    device-kernel name contains it count (a summary taken before the dispatcher changed kernels answers None, not a
    stale number). -> (value | None, file name | None)"""
    import glob
    import json
    import os
    files = sorted(glob.glob(os.path.join(profiles_dir, "r*_%s.json" % stem)))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        old = key.replace("l2_fabric_", "hbm_")  # summaries written before round 4 call the same numbers hbm_*
        vals = [e.get(key, e.get(old)) for k, e in d.items() if isinstance(e, dict) and (key in e or old in e) and kernel_sub in k]
        return (vals[-1], os.path.basename(files[-1])) if vals else (None, None)
    except Exception:
        return None, None


def collect_pmc_here(root, timeout_s=45.0):
    """Counters from THIS box for the headline kernel: the production 4096^3 HGEMM re-run for a dozen launches in rocprofv3
    child processes, one per counter set -- FETCH_SIZE and WRITE_SIZE cannot share a pass, PMC never together with the sys / hip / hsa trace
    domains (MI355X_MICROARCH.md, HBM / rocprofv3 section). Returns {"traffic": bytes per launch, "mfma_busy": fraction, "status": text,
    "seconds": wall} with None for what could not be had; never raises (the caller falls back to the committed pass and says so)."""
    import csv
    import glob
    import os
    import shutil
    import signal
    import subprocess
    import tempfile
    t0 = time.time()
    out = {"traffic": None, "read_bytes": None, "write_bytes": None, "mfma_busy": None, "status": "ok", "seconds": 0.0}
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        out["status"] = "rocprofv3 not found"
        return out
    target = os.path.join(root, "cuda-learn-notes_amd", "tools", "prof_target.py")
    # kind 14, variant 203 = the production schedule with the production epilogue (tools/profile_round.sh)
    args = [sys_executable(), target, "hgemm", "14", "0", "1", "64", "203", "4096", "12"]
    tmp = tempfile.mkdtemp(prefix="cln_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for tag, ctrs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])):
            left = timeout_s - (time.time() - t0)
            if left < 5.0:
                out["status"] = "time budget spent before pass %s" % tag
                break
            d = os.path.join(tmp, tag)
            cmd = [exe, "--kernel-trace", "--output-format", "csv", "--pmc"] + ctrs + ["-d", d, "-o", "pmc", "--"] + args
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=left)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)  # exactly the process group started here
                p.wait()
                out["status"] = "pass %s timed out" % tag
                break
            if p.returncode != 0:
                out["status"] = "pass %s: rocprofv3 exit %d" % (tag, p.returncode)
                break
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "hgemm_w4_kernel" in r["Kernel_Name"]:
                        per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for c, v in per.items():
                v = v[1:] if len(v) > 2 else v
                vals[c] = sum(v) / len(v)
        if "FETCH_SIZE" in vals:
            out["read_bytes"] = 2.0 * vals["FETCH_SIZE"] * 1024  # gfx950: FETCH_SIZE (KiB) reads half of a wide coalesced stream (tools/pmc_summary.py)
        if "WRITE_SIZE" in vals:
            out["write_bytes"] = vals["WRITE_SIZE"] * 1024
        if out["read_bytes"] is not None and out["write_bytes"] is not None:
            out["traffic"] = out["read_bytes"] + out["write_bytes"]
        if vals.get("GRBM_GUI_ACTIVE"):
            out["mfma_busy"] = (vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0) / (vals["GRBM_GUI_ACTIVE"] / 8.0)
        if out["status"] == "ok" and (out["traffic"] is None or out["mfma_busy"] is None):
            out["status"] = "counters missing from the rocprofv3 output"
    except Exception as e:  # noqa: BLE001 -- a side measurement
        out["status"] = "error: %s" % str(e)[:120]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        out["seconds"] = round(time.time() - t0, 1)
    return out


def sys_executable():
    import sys
    return sys.executable or "python3"
