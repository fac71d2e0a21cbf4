"""GPU probe: the fp32-scaled-score form of the D = 64 / 128 attention kernels (M16X_FSCALE) against the shipped pre-scaled-Q form:
TFLOPS and max |O - fp64| on N(0,1) inputs and on the amplified-key inputs of the rescale-regime tests.  python fa_fscale_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host  # noqa: E402

dev = torch.device("cuda:0")


def ref64(q, k, v):
    B, H, N, D = q.shape
    out = torch.empty(B, H, N, D, dtype=torch.float64, device=dev)
    for b in range(B):
        for h in range(H):
            s = (q[b, h].double() @ k[b, h].double().t()) / (D ** 0.5)
            out[b, h] = torch.softmax(s, dim=-1) @ v[b, h].double()
    return out


MS = "row sums on the matrix pipe"
for shape, codes in (((4, 8, 2048, 64), ((853, "pre-scaled Q (shipped)"), (990, "fp32-scaled scores"), (994, MS), (995, MS + " + fp32 scale"),
                                          (999, "fp32-scaled, 3 blocks deferred"), (997, "fp32-scaled, 5 blocks deferred"), (998, "fp32-scaled, 6 blocks deferred"))),
                     ((4, 8, 2048, 128), ((853, "pre-scaled Q (shipped)"), (990, "fp32-scaled scores"), (994, MS), (995, MS + " + fp32 scale"),
                                           (999, "fp32-scaled, 3 blocks deferred"), (997, "fp32-scaled, 5 blocks deferred"), (998, "fp32-scaled, 6 blocks deferred"))),
                     ((2, 24, 4096, 64), ((853, "pre-scaled Q (shipped)"), (994, MS))),
                     ((1, 48, 8192, 64), ((925, "pre-scaled Q (shipped)"), (992, "fp32-scaled scores"), (996, MS)))):
    B, H, N, D = shape
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    qa, ka = q.clone(), k.clone()  # amplified keys: the rescale-regime inputs of tests/test_gpu_flash_attn.py
    ka[0, 0, 900] = qa[0, 0, 5] * 3.0
    ka[0, 1, 10] = qa[0, 1, 300] * 5.0
    ka[0, H - 1, 1000] = qa[0, H - 1, 1023] * 4.0
    ramp = torch.linspace(0.2, 1.6, N, device=dev).view(1, 1, N, 1)
    ka[:, 2] = (ka[:, 2].float() * ramp[:, 0]).half()
    r_rand = ref64(q[:, :2], k[:, :2], v[:, :2]) if N <= 2048 else None
    r_amp = ref64(qa, ka, v) if N <= 2048 else None
    o = torch.zeros_like(q)
    for code, tag in codes:
        call = lambda: host.fa2_variant((8, 0, 0, code), q, k, v, o)
        bu.prewarm(call, 0.25)
        ms = bu.time_region_events(call, 100 if N <= 2048 else 25)
        torch.cuda.synchronize()
        e1 = (o[:, :2].double() - r_rand).abs().max().item() if r_rand is not None else float("nan")
        host.fa2_variant((8, 0, 0, code), qa, ka, v, o)
        torch.cuda.synchronize()
        e2 = (o.double() - r_amp).abs().max().item() if r_amp is not None else float("nan")
        print("FSCALE %-20s %-42s %8.4f ms %7.1f TF  max|O - fp64|: N(0,1) %.2e, amplified keys %.2e" %
              (shape, tag, ms, bu.mha_flops_conventional(*shape) / ms * 1e-9, e1, e2), flush=True)
