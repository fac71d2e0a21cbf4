"""Launch loop for ONE `rocprofv3 --kernel-trace` process: every attention shape of the bench line (and the larger side
rows), each pre-warmed for >= 0.3 s and then launched LAUNCHES times back to back through the C-ABI name the bench uses, so
that 4*B*H*N^2*D / (average traced duration) reproduces `roofline_fa2_*.frac` of bench.py (the round-2 trace
timed 20 cold launches and read 10 % below the bench line). fa_trace_summary.py slices the trace with the order file.
  FA_TRACE_ORDER=<json> python fa_trace_target.py [launches]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
fa = pkg.flash_attn_lib()
dev = torch.device("cuda:0")
LAUNCHES = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SHAPES = [(4, 8, 2048, 64), (4, 8, 2048, 128), (1, 48, 8192, 64), (2, 32, 4096, 128), (4, 8, 2048, 256), (2, 32, 4096, 256),
          (1, 32, 4096, 512), (1, 16, 4096, 640), (1, 16, 4096, 768), (1, 16, 4096, 1024)]
order = []
for (B, H, N, D) in SHAPES:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    fn = fa.flash_attn_mma_stages_split_q_shared_qkv if D <= 256 else fa.flash_attn_mma_stages_split_q_tiling_qkv
    torch.cuda.synchronize()
    warm, t0 = 0, time.time()
    while time.time() - t0 < 0.3:
        for _ in range(20):
            fn(q, k, v, o, 2)
        warm += 20
        torch.cuda.synchronize()
    for _ in range(LAUNCHES):
        fn(q, k, v, o, 2)
    torch.cuda.synchronize()
    order.append({"shape": [B, H, N, D], "warm": warm, "launches": LAUNCHES,
                  "describe": pkg.manifest.describe(fn.__name__, (B, H, N, D), 2)})
    del q, k, v, o
json.dump(order, open(os.environ.get("FA_TRACE_ORDER", os.path.join(ROOT, "gpurun_out", "fa_trace_order.json")), "w"), indent=1)
