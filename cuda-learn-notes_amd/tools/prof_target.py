"""Small launch loop for rocprofv3: run ONE kernel variant `iters` times so the trace/PMC output holds
only that kernel.   python prof_target.py hgemm <kind> <layout> <tile> <bk> <stages> [size] [iters]
                    python prof_target.py fa <B> <H> <N> <D> <stages> [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    pkg = entry.load_package()
    from cuda_learn_notes_amd import bench_utils as bu, host
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    what = sys.argv[1]
    if what == "hgemm":
        kind, layout, tile, bk, stages = map(int, sys.argv[2:7])
        S = int(sys.argv[7]) if len(sys.argv) > 7 else 4096
        iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
        a = torch.randn(S, S, dtype=torch.half, device=dev)
        b = torch.randn(S, S, dtype=torch.half, device=dev)
        bb = bu.as_col_major(b) if layout else b
        c = torch.zeros(S, S, dtype=torch.half, device=dev)
        stride = bu.make_block_swizzle_stride(S, S)
        for _ in range(iters):
            host.hgemm_variant(kind, layout, tile, bk, stages, a, bb, c, 1, stride)
    elif what == "fa":
        B, H, N, D, stages = map(int, sys.argv[2:7])
        iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
        fa = pkg.flash_attn_lib()
        q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
        o = torch.zeros_like(q)
        fn = fa.flash_attn_mma_stages_split_q_shared_qkv if D <= 256 else fa.flash_attn_mma_stages_split_q_tiling_qkv
        for _ in range(iters):
            fn(q, k, v, o, stages)
    elif what == "fa2":  # fa2 <D> <nw> <opt> <abl> <B> <H> <N> [iters]
        D, nw, opt, abl, B, H, N = map(int, sys.argv[2:9])
        iters = int(sys.argv[9]) if len(sys.argv) > 9 else 10
        q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
        o = torch.zeros_like(q)
        for _ in range(iters):
            host.fa2_variant((nw, 0, opt, abl), q, k, v, o)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
