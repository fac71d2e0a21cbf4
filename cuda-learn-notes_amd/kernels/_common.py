"""Shared plumbing of the re-authored bench scripts (cuda-learn-notes_amd/kernels/<topic>/<topic>.py).

Each script keeps the reference script's CLI, row tags and printed columns but is
table-driven: a list of (tag, function name, arguments) rows walked by one runner. With no GPU present
only the stock-torch rows run, on CPU -- BASELINE config C1 (the reference's own CPU-runnable path); the
kernel rows are reported as skipped because the HIP path has no CPU fallback.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

torch.set_grad_enabled(False)
HAS_GPU = torch.cuda.is_available()
DEVICE = torch.device("cuda:0") if HAS_GPU else torch.device("cpu")


PKG = entry.load_package()  # registers the package as `cuda_learn_notes_amd` (no .so is loaded yet)


def package():
    return PKG


def sync():
    if HAS_GPU:
        torch.cuda.synchronize()


def timed(call, warmup, iters):
    """Reference protocol (e.g. kernels/elementwise/elementwise.py:25-56): warmup, sync, time.time() around
    `iters` calls, sync. Returns (last result, mean ms)."""
    out = None
    for _ in range(warmup):
        out = call()
    sync()
    t0 = time.time()
    for _ in range(iters):
        out = call()
    sync()
    return out, (time.time() - t0) * 1000.0 / iters


def emit_json(rows, path_env="CLN_AMD_BENCH_JSON"):
    """Optional machine-readable copy of the table (one JSON object per row) next to the human output."""
    path = os.environ.get(path_env)
    if path:
        with open(path, "a") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


def hbm_row(tag, shape, ms, nbytes):
    from cuda_learn_notes_amd.bench_utils import PEAK_HBM_GBPS
    gbps = nbytes / ms * 1e-6
    return {"kernel": tag, "shape": list(shape), "ms": ms, "gbps": gbps,
            "roofline": {"bound": "hbm", "peak": PEAK_HBM_GBPS, "achieved": gbps, "frac": gbps / PEAK_HBM_GBPS}}


def run_table(title_width, sections, out_width=20, json_rows=None, bytes_per_elem=None):
    """Walk a list of (header, [(tag, callable | None, out_tensor | None)]) sections and print the reference
    scripts' row format `out_<tag>: [v0, v1, v2], time:<ms>ms`. A row whose callable is None is a kernel row
    without a GPU: it is reported as skipped (no CPU fallback)."""
    for header, rows, warmup, iters in sections:
        print("-" * title_width)
        print(" " * (title_width // 2 - 5) + header)
        print("-" * title_width)
        for tag, call, out, shape, nbytes in rows:
            info = "out_" + tag
            if call is None:
                print(f"{info:>{out_width}}: skipped (no GPU: the HIP kernel path has no CPU fallback)")
                continue
            if out is not None:
                out.fill_(0)
            res, ms = timed(call, warmup, iters)
            val = out if out is not None else res
            vals = [f"{round(v, 8):<12}" for v in val.flatten()[:3].float().cpu().tolist()]
            print(f"{info:>{out_width}}: {vals}, time:{ms:.8f}ms")
            if json_rows is not None and nbytes:
                json_rows.append(hbm_row(tag, shape, ms, nbytes))
    print("-" * title_width)
