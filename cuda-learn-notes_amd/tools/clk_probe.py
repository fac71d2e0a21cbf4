"""Sample rocm-smi clocks/power while a kernel loops (DVFS view of a variant). python clk_probe.py"""
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host  # noqa: E402

dev = torch.device("cuda:0")
hg = pkg.hgemm_lib()
hg.init_cublas_handle()
S = 4096
a = torch.randn(S, S, dtype=torch.half, device=dev)
b = torch.randn(S, S, dtype=torch.half, device=dev)
bt = bu.as_col_major(b)
c = torch.zeros(S, S, dtype=torch.half, device=dev)
z = torch.zeros(S, S, dtype=torch.half, device=dev)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True,
                             timeout=20).stdout
        return out.strip()[:600]
    except Exception as e:
        return "smi error %s" % e


w4 = lambda v, x=None, y=None, lay=0: (lambda: host.hgemm_variant(14, lay, 1, 64, v, x if x is not None else a, y if y is not None else (bt if lay else b), c, 1, 2048))
if os.environ.get("CLK_CSTORE"):  # the C-store forms of the production schedule: plain (26), non-temporal (203 = what ships), write-through (204)
    cands = [("w4 26 plain C stores NN", w4(26)), ("w4 26 nt C stores NN", w4(203)), ("w4 26 sc0sc1 C stores NN", w4(204)),
             ("w4 26 plain C stores TN", w4(26, lay=1)), ("w4 26 nt C stores TN", w4(203, lay=1)), ("rocblas TN", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c))]
else:
  cands = [("w4 v4 NN", w4(4)), ("w4 v4 snake", w4(20)), ("w4 v4 B-major", w4(36)), ("w4 v4 B-major snake", w4(52)),
         ("w4 v4 TN", w4(4, lay=1)), ("w4 v4 TN snake", w4(20, lay=1)), ("w4 v4 TN B-major", w4(36, lay=1)), ("w4 v4 TN B-major snake", w4(52, lay=1)),
         ("w4 mfma-only", w4(117)), ("w4 mfma-only snake", w4(120)), ("w4 mfma-only B-major", w4(136)), ("w4 mfma-only B-m snake", w4(152)),
         ("rocblas TN", lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c)),
         ("rocblas NN", lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c)),
         ("pp16 split NN", lambda: host.hgemm_variant(8, 0, 1, 64, 4, a, b, c, 1, 2048)),
         ("pp16 mfma-only", lambda: host.hgemm_variant(7, 0, 1, 64, 7, a, b, c, 1, 2048)),
         ("m32 mfma-only", lambda: host.hgemm_variant(10, 0, 1, 64, 23, a, b, c, 1, 2048)),
         ("pp16 split zeros", lambda: host.hgemm_variant(8, 0, 1, 64, 4, z, z, c, 1, 2048))]
print(smi(), flush=True)
for tag, fn in cands:
    res = {}

    def sampler():
        time.sleep(1.0)
        res["smi"] = smi()

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.5:
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        n += 200
    dt = time.time() - t0
    th.join()
    print("%-24s %7.1f TF sustained | %s" % (tag, 2.0 * S ** 3 * n / dt * 1e-12, res.get("smi")), flush=True)
