"""Round-5 probe of the one-wave-per-SIMD kernel for head dims 640 / 768 / 1024 (csrc/flash_attn_dw4.cuh) against the production
planner's kernel (round 4: flash_attn_dring.cuh): max error vs fp32 attention, then TFLOPS (4 B H N^2 D) over one event-timed region.
Variant codes: csrc/probe/flash_attn_probe.hip (1300 + opt, 1400 + 10 * KPF + VPF)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, host  # noqa: E402

dev = torch.device("cuda:0")
fa = pkg.flash_attn_lib()
prod = fa.flash_attn_mma_stages_split_q_tiling_qkv
quick = "--quick" in sys.argv


def ref_attn(q, k, v):
    return F.scaled_dot_product_attention(q.float(), k.float(), v.float())


def check(tag, fn, o, ref):
    o.zero_()
    try:
        fn()
        torch.cuda.synchronize()
        err = (o.float() - ref).abs().max().item()
        bad = not torch.isfinite(o.float()).all().item()
        print("CHK %-34s max|err| %.3e%s" % (tag, err, "  NON-FINITE" if bad else ""), flush=True)
    except Exception as e:  # noqa: BLE001
        print("CHK %-34s ERR %s" % (tag, str(e)[:100]), flush=True)


# ---- correctness: random inputs, and keys whose scale GROWS along the sequence (the running maximum rises by > 8 in the log2 domain
# many times: the deferred form must take its rescale path)
for (B, H, N, D) in [(1, 2, 512, 512), (1, 3, 1152, 512), (1, 2, 512, 640), (1, 2, 64, 640), (1, 3, 1088, 768), (1, 2, 512, 768), (1, 2, 512, 1024), (1, 8, 2048, 1024)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    for name, kk in (("randn", k), ("growing keys", k * torch.linspace(0.2, 6.0, N, device=dev).view(1, 1, N, 1).half())):
        ref = ref_attn(q, kk, v)
        check("%s %s production" % ((B, H, N, D), name), lambda: prod(q, kk, v, o, 2), o, ref)
        for abl in (1300, 1301, 1302, 1316, 1332, 1364, 1348, 1396, 1412, 1413, 1414) + ((1600, 1601, 1602) if D in (640, 768) else ()):
            check("%s %s dw4 %d" % ((B, H, N, D), name, abl), lambda: host.fa2_variant((4, 0, 0, abl), q, kk, v, o), o, ref)
    # bit-identity of the `stages = 1` form
    o1, o2 = torch.zeros_like(q), torch.zeros_like(q)
    host.fa2_variant((4, 0, 0, 1300), q, k, v, o1)
    host.fa2_variant((4, 0, 0, 1301), q, k, v, o2)
    o3, o4, o5 = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(q)
    host.fa2_variant((4, 0, 0, 1412), q, k, v, o3)
    host.fa2_variant((4, 0, 0, 1413), q, k, v, o4)
    prod(q, k, v, o5, 1)
    if D >= 640:  # two tiles per loop iteration: same arithmetic, same bits
        u = {640: 1444, 768: 1540, 1024: 1540}[D]
        o8, o9 = torch.zeros_like(q), torch.zeros_like(q)
        host.fa2_variant((4, 0, 0, u), q, k, v, o8)
        host.fa2_variant((4, 0, 0, u + 1), q, k, v, o9)
        torch.cuda.synchronize()
        print("BIT %s unrolled-by-2 form == plain: %s; its stages 1 == 2: %s" % ((B, H, N, D), torch.equal(o8, o1), torch.equal(o9, o8)), flush=True)
    if D in (640, 768):  # the one-barrier-per-tile form: same arithmetic, same bits
        o6, o7 = torch.zeros_like(q), torch.zeros_like(q)
        host.fa2_variant((4, 0, 0, 1600), q, k, v, o6)
        host.fa2_variant((4, 0, 0, 1601), q, k, v, o7)
        torch.cuda.synchronize()
        print("BIT %s one-barrier form == two-barrier form: %s; its stages 1 == 2: %s" % ((B, H, N, D), torch.equal(o6, o1), torch.equal(o7, o6)), flush=True)
    torch.cuda.synchronize()
    print("BIT %s stages 1 == 2: %s; production options (carry / M0 walk / spread) == plain: %s; their stages 1 == 2: %s; production name stages 1: %s"
          % ((B, H, N, D), torch.equal(o1, o2), torch.equal(o3, o1), torch.equal(o4, o3), torch.equal(o5, o3)), flush=True)

# ---- timing
shapes = [(1, 32, 4096, 512), (1, 16, 4096, 640), (1, 16, 4096, 768), (1, 16, 4096, 1024)] + ([] if quick else [(1, 8, 8192, 1024), (2, 16, 2048, 768), (4, 8, 2048, 512)])
for (B, H, N, D) in shapes:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    fl = bu.mha_flops_conventional(B, H, N, D)
    cands = [("production stages 2", lambda: prod(q, k, v, o, 2)), ("production stages 1", lambda: prod(q, k, v, o, 1))]
    codes = [1300, 1301, 1316, 1332, 1364, 1348, 1396, 1412, 1413]
    if D in (768, 1024):
        codes += [1304, 1416]
    codes += {1024: [1511, 1533, 1542], 768: [1511, 1533], 640: [1511, 1533], 512: [1304, 1416, 1511, 1544, 1542]}[D]
    if D == 512:
        codes = [c for c in codes if c not in (1332, 1364, 1396)]
    if D in (640, 768):
        codes += [1600, 1601, 1604]
    codes += {640: [1444, 1445, 1540, 1428], 768: [1540, 1541, 1476, 1428], 1024: [1540, 1541, 1476, 1428]}.get(D, [])  # + 128: two tiles per iteration
    for abl in codes:
        cands.append(("dw4 %d" % abl, (lambda a: lambda: host.fa2_variant((4, 0, 0, a), q, k, v, o))(abl)))
    for rnd in range(2):
        for tag, fn in cands:
            try:
                bu.prewarm(fn, 0.15)
                ms = bu.time_region_events(fn, 20)
                print("FA %s r%d %-22s %8.4f ms %7.1f TF" % ((B, H, N, D), rnd, tag, ms, fl / ms * 1e-9), flush=True)
            except Exception as e:  # noqa: BLE001
                print("FA %s %s ERR %s" % ((B, H, N, D), tag, str(e)[:100]), flush=True)
