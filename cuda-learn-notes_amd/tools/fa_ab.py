"""Same-box A/B of two builds of libcln_amd.so on the attention shapes of the bench line (boxes differ by +-4 %, so a kernel change is only
measurable against its predecessor on the box it runs on): both libraries are loaded through ctypes, every shape is timed in ALTERNATING blocks
(A, B, A, B, ... of `launches` back-to-back launches each, HIP-event-timed on the launch stream), the per-launch medians over the blocks are
compared, and the two outputs are compared bit for bit.
    python tools/fa_ab.py <baseline.so> <candidate.so> [launches per block = 100] [blocks = 7]"""
import ctypes
import os
import statistics
import sys

import torch

SHAPES = [(4, 8, 2048, 64), (4, 8, 2048, 128), (1, 48, 8192, 64), (2, 32, 4096, 128), (4, 8, 2048, 256), (2, 32, 4096, 256),
          (1, 32, 4096, 512), (2, 16, 2048, 320), (1, 16, 4096, 640), (1, 16, 4096, 768), (1, 16, 4096, 1024)]


def entry(lib, D):
    fn = lib.flash_attn_mma_stages_split_q_shared_qkv if D <= 256 else lib.flash_attn_mma_stages_split_q_tiling_qkv
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    fn.restype = ctypes.c_int
    return fn


def main():
    a_path, b_path = sys.argv[1], sys.argv[2]
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    libs = [ctypes.CDLL(os.path.abspath(a_path)), ctypes.CDLL(os.path.abspath(b_path))]
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    print("FAAB baseline %s  candidate %s  (%d launches per block, %d blocks each, alternating)" % (a_path, b_path, launches, blocks))
    for (B, H, N, D) in SHAPES:
        torch.manual_seed(0)
        q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
        outs = [torch.zeros_like(q), torch.zeros_like(q)]
        fns = [entry(lib, D) for lib in libs]
        call = [lambda i=i: fns[i](q.data_ptr(), k.data_ptr(), v.data_ptr(), outs[i].data_ptr(), B, H, N, D, 2, st) for i in range(2)]
        if call[0]() != 0 or call[1]() != 0:
            print("FAAB %s: unsupported" % ((B, H, N, D),))
            continue
        torch.cuda.synchronize()
        same = bool(torch.equal(outs[0], outs[1]))
        for _ in range(3):  # warm both
            for i in range(2):
                for _ in range(launches):
                    call[i]()
        torch.cuda.synchronize()
        t = [[], []]
        for _ in range(blocks):
            for i in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(launches):
                    call[i]()
                e1.record()
                e1.synchronize()
                t[i].append(e0.elapsed_time(e1) * 1e3 / launches)
        ma, mb = statistics.median(t[0]), statistics.median(t[1])
        fl = 4.0 * B * H * N * N * D
        print("FAAB [%d,%d,%d,%d]  baseline %8.2f us %7.1f TF   candidate %8.2f us %7.1f TF   candidate/baseline speed %.3f   bit-identical %s"
              % (B, H, N, D, ma, fl / ma * 1e-6, mb, fl / mb * 1e-6, ma / mb, same), flush=True)


if __name__ == "__main__":
    main()
