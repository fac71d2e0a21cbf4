"""GPU probe: the interleaved-chunk walk of the one-wave-per-SIMD HGEMM (tile_coords_interleaved) against the per-XCD-range walk at the large sizes, in two
processes per size ($CLN_AMD_W4_BLOCK_WALK = 0 / 1 is read once per process).   python hg_block_walk_probe.py [child size walk]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import __graft_entry__ as entry
    pkg = entry.load_package()
    from cuda_learn_notes_amd import bench_utils as bu
    S, walk = int(sys.argv[2]), sys.argv[3]
    dev = torch.device("cuda:0")
    hg = pkg.hgemm_lib()
    torch.manual_seed(1)
    a = torch.randn(S, S, dtype=torch.half, device=dev)
    b = torch.randn(S, S, dtype=torch.half, device=dev)
    bt = bu.as_col_major(b)
    c = torch.zeros(S, S, dtype=torch.half, device=dev)
    stride = int(os.environ.get("STRIDE", "0")) or bu.make_block_swizzle_stride(S, S)
    fl = bu.hgemm_flops(S, S, S)
    import hashlib
    for tag, fn, bb, st in (("NN stages=2", hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem, b, 2), ("TN stages=2", hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4, bt, 2),
                            ("NN stages=3", hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem, b, 3)):
        call = lambda: fn(a, bb, c, st, True, stride)
        bu.prewarm(call, 0.4)
        ms = bu.time_region_events(call, max(8, int(150 * (4096.0 / S) ** 3)))
        torch.cuda.synchronize()
        h = hashlib.sha1(c.cpu().numpy().tobytes()).hexdigest()[:12]
        print("WALK %6d %-12s interleaved walk %s stride %5d  %9.3f ms %7.1f TF  C sha1 %s" % (S, tag, walk, stride, ms, fl / ms * 1e-9, h), flush=True)
    sys.exit(0)

for S, walk, stride in [(S, w, "0") for S in (8192, 12544, 15360, 16384, 10240) for w in ("0", "1")] + \
        [(S, "1", st) for S in (12544, 15360, 16384) for st in ("1024", "2048", "4096")]:
    if True:
        env = dict(os.environ, CLN_AMD_W4_BLOCK_WALK=walk, STRIDE=stride)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(S), walk], env=env, capture_output=True, text=True)
        sys.stdout.write("".join(l + "\n" for l in r.stdout.splitlines() if l.startswith("WALK")))
        if r.returncode != 0:
            sys.stdout.write("WALK child failed: %s\n" % r.stderr[-300:])
        sys.stdout.flush()
