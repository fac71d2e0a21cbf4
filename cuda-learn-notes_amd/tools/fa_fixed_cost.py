"""Fixed cost of one attention workgroup: 256 workgroups of 256 query rows (one per CU), KV length swept, so
time = fixed + slope * N. Production path (ping-pong kernel) at D = 64 / 128."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu
dev = torch.device("cuda:0")
fa = pkg.flash_attn_lib()
for D in (64, 128):
    pts = []
    for N in (512, 1024, 2048, 4096, 8192):
        H = 256 * 256 // N
        q, k, v = (torch.randn(1, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
        o = torch.zeros_like(q)
        fn = lambda: fa.flash_attn_mma_stages_split_q_shared_qkv(q, k, v, o, 2)
        ms = min(bu.time_call_events(fn, 5, 40)[0], bu.time_call_events(fn, 2, 40)[0])
        pts.append((N, ms * 1e3))
        print("D=%d H=%4d N=%5d  %8.2f us  %7.1f TF" % (D, H, N, ms * 1e3, bu.mha_flops_conventional(1, H, N, D) / ms * 1e-9), flush=True)
    (n0, t0), (n1, t1) = pts[1], pts[-1]
    slope = (t1 - t0) / (n1 - n0)
    print("D=%d: slope %.3f us per 1024 keys, fixed %.2f us (of %.2f us at N = 2048)" % (D, slope * 1024, t0 - slope * n0, pts[2][1]), flush=True)
