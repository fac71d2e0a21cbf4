"""Round-5 probe: rows per workgroup of the row kernels (csrc/rowwise.cuh rows_per_wg; $CLN_AMD_ROWS_PER_WG is read once per process) -- the bench's own
bandwidth rows (rotating buffer sets, launch-inclusive event regions; GPU rows only: no CPU leg, no checker) for the softmax / layer-norm / rms-norm rungs.  Run once per setting."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
import bench_configs as bc  # noqa: E402

dev = torch.device("cuda:0")
tag = os.environ.get("CLN_AMD_ROWS_PER_WG", "default")
rows = bc.bandwidth_rows(dev, None, shapes=((4096, 4096), (8192, 8192), (4096, 2048)), cpu_shape=(0, 0))
for r in rows:
    if any(k in r["kernel"] for k in ("softmax", "norm", "reduce")) or os.environ.get("BW_ALL"):
        print("RPW=%-7s %-40s %-12s %s %8.2f us %7.1f GB/s (same buffers %7.1f; the kernel on 64 rows: %s us)" % (tag, r["kernel"], r["shape"], r["dtype"], r["us_per_launch"], r["gbps"],
                                                                                   r["gbps_same_buffers"], r.get("us_64_row_launch")), flush=True)
