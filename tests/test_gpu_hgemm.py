"""GPU parity: every HGEMM entry point through the C-ABI vs the CPU oracle (fp32-accumulate product of
the same fp16 inputs). Tolerance: the kernels accumulate in fp32 and round once to fp16, so the result
must be within one fp16 ulp of the fp32 truth: rtol 2^-10 (plus atol 2e-3 for near-cancelled sums)."""
import re

import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2 ** -10, 2e-3


def seeded(seed, *shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).half()


@pytest.fixture(scope="module")
def hg(built, dev):
    lib = built.hgemm_lib()
    lib.init_cublas_handle()
    yield lib
    lib.destroy_cublas_handle()


def check(c, a, b):
    truth = a.float() @ b.float()
    got = c.cpu().float()
    err = (got - truth).abs()
    bound = ATOL + RTOL * truth.abs()
    bad = (err > bound).sum().item()
    assert bad == 0, "mismatches=%d max_err=%g" % (bad, err.max().item())


G3_NN = [e for e in [
    "hgemm_naive_f16", "hgemm_sliced_k_f16", "hgemm_t_8x8_sliced_k_f16x4", "hgemm_t_8x8_sliced_k_f16x4_pack",
    "hgemm_t_8x8_sliced_k_f16x4_bcf", "hgemm_t_8x8_sliced_k_f16x4_pack_bcf", "hgemm_t_8x8_sliced_k_f16x8_pack_bcf",
    "hgemm_t_8x8_sliced_k_f16x8_pack_bcf_dbuf", "hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf",
    "hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf_async", "hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf",
    "hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf_async", "hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf",
    "hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf_async", "hgemm_cublas_tensor_op_nn", "hgemm_wmma_m16n16k16_naive",
    "hgemm_wmma_m16n16k16_mma4x2", "hgemm_wmma_m16n16k16_mma4x2_warp2x4",
    "hgemm_wmma_m16n16k16_mma4x2_warp2x4_dbuf_async", "hgemm_wmma_m32n8k16_mma2x4_warp2x4_dbuf_async",
    "hgemm_mma_m16n8k16_naive", "hgemm_mma_m16n8k16_mma2x4_warp4x4"]]
G6_NN = ["hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages", "hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem",
         "hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem", "hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem",
         "hgemm_mma_m16n8k16_mma2x4_warp4x4_stages", "hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem",
         "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_x4",
         "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_rr",
         "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_swizzle"]
G6_TN = ["hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem_tn",
         "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4", "hgemm_mma_stages_block_swizzle_tn_cute"]


@pytest.mark.parametrize("name", G3_NN)
def test_g3_functions(hg, dev, name):
    M, N, K = 256, 384, 512  # asymmetric on purpose
    a, b = seeded(1, M, K), seeded(2, K, N)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    getattr(hg, name)(a.to(dev), b.to(dev), c)
    check(c, a, b)


@pytest.mark.parametrize("stages", [2, 3, 4, 5])
@pytest.mark.parametrize("swizzle", [False, True])
@pytest.mark.parametrize("name", G6_NN)
def test_g6_nn_functions(hg, dev, name, stages, swizzle):
    M, N, K = 512, 768, 320  # K % 64 == 0; 5 K-tiles of 64
    a, b = seeded(3, M, K), seeded(4, K, N)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    getattr(hg, name)(a.to(dev), b.to(dev), c, stages, swizzle, 256)
    check(c, a, b)


@pytest.mark.parametrize("stages", [2, 3, 4])
@pytest.mark.parametrize("swizzle", [False, True])
@pytest.mark.parametrize("name", G6_TN + ["hgemm_cublas_tensor_op_tn"])
def test_tn_functions(hg, dev, built, name, stages, swizzle):
    from cuda_learn_notes_amd.bench_utils import as_col_major
    M, N, K = 512, 768, 320
    a, b = seeded(5, M, K), seeded(6, K, N)
    bt = as_col_major(b)  # [K,N] shape, [N,K] storage
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    if name.startswith("hgemm_cublas"):
        getattr(hg, name)(a.to(dev), bt.to(dev), c)
    else:
        getattr(hg, name)(a.to(dev), bt.to(dev), c, stages, swizzle, 256)
    check(c, a, b)


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 6, 7, 8])
@pytest.mark.parametrize("bk,stages", [(64, 2), (64, 3), (64, 5), (32, 2), (32, 3), (32, 4), (32, 5)])
def test_every_ring_instantiation(built, dev, layout, tile, bk, stages):
    """Each (tile, BK, stages, layout) template instantiation, incl. K tiles fewer than stages (tiles 6 / 7 / 8 = 64x128,
    64x64 on four and on two waves: the small-problem tiles; 64-column NN images have their own bank swizzle)."""
    from cuda_learn_notes_amd import host
    from cuda_learn_notes_amd.bench_utils import as_col_major
    lds = stages * {0: 256, 1: 512, 2: 384, 3: 384, 6: 192, 7: 128, 8: 128}[tile] * bk * 2
    if lds > 160 * 1024:
        pytest.skip("does not fit LDS")
    for K in (bk, 3 * bk, 9 * bk):
        M, N = 512, 512
        a, b = seeded(7 + K, M, K), seeded(8 + K, K, N)
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        bb = as_col_major(b) if layout else b
        host.hgemm_variant(0, layout, tile, bk, stages, a.to(dev), bb.to(dev), c, swizzle=1, swizzle_stride=256)
        check(c, a, b)


@pytest.mark.parametrize("layout", [0, 1])
def test_pingpong_kernel(built, dev, layout):
    """256x256x64 phase-staggered kernel (kind 3): several K depths incl. a single K tile, many launches
    (a barrier/DMA ordering slip shows up as rare wrong tiles, so repeat and compare bit-exactly)."""
    from cuda_learn_notes_amd import host
    from cuda_learn_notes_amd.bench_utils import as_col_major
    for (M, N, K) in ((256, 256, 64), (512, 768, 320), (512, 512, 128), (1024, 1024, 1024), (2048, 2048, 512)):
        a, b = seeded(70 + K, M, K), seeded(71 + K, K, N)
        bb = (as_col_major(b) if layout else b).to(dev)
        ad = a.to(dev)
        first = None
        for rep in range(10):
            c = torch.zeros(M, N, dtype=torch.half, device=dev)
            kind, st = ((3, 2), (5, 8), (5, 4), (8, 4), (9, 4))[rep % 5]  # plain / LDS-epilogue 8 slots / 4 slots / split DMA / k-half ring
            host.hgemm_variant(kind, layout, 1, 64, st, ad, bb, c, swizzle=rep & 1, swizzle_stride=512)
            if first is None:
                check(c, a, b)
                first = c
            else:
                assert torch.equal(c, first)


def test_identity_times_asymmetric_b_is_exact(hg, dev):
    """A = I catches any row/column transposition in fragment or C layouts (cdna guide G9)."""
    n = 512
    a = torch.eye(n).half()
    b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 2039 - 1000).half() / 8  # asymmetric, exact in fp16
    for name in ("hgemm_mma_m16n8k16_naive", "hgemm_mma_m16n8k16_mma2x4_warp4x4"):
        c = torch.zeros(n, n, dtype=torch.half, device=dev)
        getattr(hg, name)(a.to(dev), b.to(dev), c)
        assert torch.equal(c.cpu(), b), name
    for name in G6_NN:
        c = torch.zeros(n, n, dtype=torch.half, device=dev)
        getattr(hg, name)(a.to(dev), b.to(dev), c, 2, True, 256)
        assert torch.equal(c.cpu(), b), name
    # and B = I: C must equal A
    a2 = b.t().contiguous()
    c = torch.zeros(n, n, dtype=torch.half, device=dev)
    hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem(a2.to(dev), torch.eye(n).half().to(dev), c, 3, False, 1)
    assert torch.equal(c.cpu(), a2)


def test_block_swizzle_is_a_pure_schedule_change(hg, dev):
    M = N = K = 1024
    a, b = seeded(9, M, K).to(dev), seeded(10, K, N).to(dev)
    outs = []
    for swizzle, stride in ((False, 1), (True, 256), (True, 512), (True, 2048)):
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem(a, b, c, 2, swizzle, stride)
        outs.append(c)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_k_not_multiple_of_64_uses_bk32(hg, dev):
    M, N, K = 256, 256, 96
    a, b = seeded(11, M, K), seeded(12, K, N)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    hg.hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem(a.to(dev), b.to(dev), c, 3, False, 1)
    check(c, a, b)


def test_unsupported_shape_raises(hg, dev):
    a = torch.zeros(100, 64, dtype=torch.half, device=dev)
    b = torch.zeros(64, 100, dtype=torch.half, device=dev)
    c = torch.zeros(100, 100, dtype=torch.half, device=dev)
    with pytest.raises(RuntimeError, match="multiples of the block tile"):
        hg.hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem(a, b, c, 2, False, 1)
    hg.hgemm_naive_f16(a, b, c)  # the naive rungs take any shape
    hg.hgemm_mma_m16n8k16_naive(a, b, c)


@pytest.mark.parametrize("size", [1024, 4096, 8192])
def test_baseline_configs_sampled_rows(hg, dev, size):
    """C2 / C3 sizes: rows sampled across the matrix vs the fp32 oracle computed on CPU."""
    from cuda_learn_notes_amd.bench_utils import make_block_swizzle_stride
    M = N = K = size
    torch.manual_seed(size)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    rows = torch.arange(0, M, max(1, M // 64))[:64]
    truth = a[rows].cpu().float() @ b.cpu().float()
    stride = make_block_swizzle_stride(N, K)
    for name, args in (("hgemm_mma_m16n8k16_mma2x4_warp4x4", ()),
                       ("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", (2, True, stride)),
                       ("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", (3, False, 1)),
                       ("hgemm_cublas_tensor_op_nn", ())):
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        getattr(hg, name)(a, b, c, *args)
        got = c[rows].cpu().float()
        err = (got - truth).abs()
        assert (err <= ATOL + RTOL * truth.abs()).all(), (name, err.max().item())


@pytest.mark.parametrize("size", [4096, 8192])
def test_baseline_configs_full_matrix(hg, dev, size):
    """BASELINE config C3 (4096^3, 8192^3): EVERY element of C against a GPU fp32 product (torch.matmul on fp32 copies of the
    operands: rocBLAS SGEMM, exact-f32 matrix instruction), which is itself pinned to the CPU fp32 oracle on sampled rows
    (the round-2 test looked at 64 sampled rows only). NN and TN rungs of the headline family."""
    from cuda_learn_notes_amd.bench_utils import as_col_major, make_block_swizzle_stride
    M = N = K = size
    torch.manual_seed(size + 1)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    truth = a.float() @ b.float()
    rows = torch.arange(0, M, M // 32)[:32]
    cpu_truth = a[rows].cpu().float() @ b.cpu().float()
    # fp32 accumulation order differs between the two fp32 products: |sum| ~ 64-90, 4096-8192 terms
    assert (truth[rows].cpu() - cpu_truth).abs().max().item() <= 2e-3
    stride = make_block_swizzle_stride(N, K)
    bt = as_col_major(b)
    for name, bb in (("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", b),
                     ("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4", bt)):
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        getattr(hg, name)(a, bb, c, 2, True, stride)
        err = (c.float() - truth).abs()
        bad = err > ATOL + RTOL * truth.abs()
        assert not bad.any().item(), (name, int(bad.sum().item()), err.max().item())
        # config C3 names stages in {2, 3, 4} (hgemm.py:359-361): the ring-of-slots form of the same kernel, the whole matrix bit-identical
        for st in (3, 4):
            cs = torch.zeros(M, N, dtype=torch.half, device=dev)
            getattr(hg, name)(a, bb, cs, st, True, stride)
            assert torch.equal(cs, c), (name, st)
    del truth


@pytest.mark.parametrize("M,N,K", [(4096, 8192, 2048), (2048, 1024, 8192), (3072, 3072, 3072), (16384, 16384, 1024),
                                   (1536, 2560, 4096), (64, 128, 64), (12800, 12800, 512), (2560, 2560, 2560),
                                   (6144, 6144, 512), (192, 256, 64), (3072, 4096, 192), (4096, 4096, 96),
                                   (3584, 3584, 1024)])
def test_rectangular_and_large_shapes(hg, built, dev, M, N, K):
    """Every tile policy branch (256x256 and 192x256 ping-pong, ring 128x256, 64x128, 128x128) on non-square and large
    problems (the reference sweeps up to M=N=12800/16384, hgemm.py MMNK): sampled rows vs the fp32 product, NN and TN
    agree. manifest.describe() says which kernel each shape runs (tests/test_describe.py pins the policy)."""
    from cuda_learn_notes_amd.bench_utils import as_col_major, make_block_swizzle_stride
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    rows = torch.unique(torch.cat([torch.arange(0, M, max(1, M // 48)), torch.tensor([M - 1])]))
    truth = a[rows].float() @ b.float()
    stride = make_block_swizzle_stride(N, K)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem(a, b, c, 2, True, stride)
    err = (c[rows].float() - truth).abs()
    assert (err <= ATOL + RTOL * truth.abs()).all(), err.max().item()
    ct = torch.zeros(M, N, dtype=torch.half, device=dev)
    hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4(a, as_col_major(b), ct, 2, True, stride)
    assert torch.equal(ct, c)  # same k order, same fp32 accumulation: TN and NN agree bit for bit


@pytest.mark.parametrize("M,N,K", [(8192, 8192, 16384), (8448, 8192, 16384)])
def test_operands_past_the_infinity_cache_take_the_interleaved_walk(hg, dev, M, N, K):
    """A + B >= 512 MB and >= 1024 tiles with block swizzle on: the XCDs take the band walk in interleaved chunks (csrc/hgemm_mfma.cuh
    tile_coords_interleaved; the sizes of the reference README, 12544^3 ... 16384^3, run it). A pure schedule change: the result equals the
    un-swizzled launch bit for bit (exact grid: 32 x 32 tiles; ragged: 33 x 32, a last round of 32 tiles), at stages 2 and 3, and sampled rows
    match the fp32 product."""
    torch.manual_seed(M + K)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    rows = torch.unique(torch.cat([torch.arange(0, M, M // 24), torch.tensor([M - 1, M - 257])]))
    truth = a[rows].float() @ b.float()
    plain = torch.zeros(M, N, dtype=torch.half, device=dev)
    hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem(a, b, plain, 2, False, 1)
    err = (plain[rows].float() - truth).abs()
    assert (err <= ATOL + RTOL * truth.abs()).all(), err.max().item()
    for stages in (2, 3):
        for stride in (2048, 1792):
            c = torch.zeros(M, N, dtype=torch.half, device=dev)
            hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem(a, b, c, stages, True, stride)
            assert torch.equal(c, plain), (stages, stride)


@pytest.mark.parametrize("layout", [0, 1])
def test_probe_kernels_stay_correct(built, dev, layout):
    """Tuning hooks kept in the library as measured (slower) alternatives must stay bit-identical to the shipped
    kernel: kind 10 = ping-pong on mfma_32x32x16, ring tile 4 = 256x256 tile with 4 waves x 128x128."""
    from cuda_learn_notes_amd import host
    from cuda_learn_notes_amd.bench_utils import as_col_major
    for (M, N, K) in ((256, 256, 64), (512, 768, 320), (1024, 1024, 1024)):
        a, b = seeded(90 + K, M, K), seeded(91 + K, K, N)
        bb = (as_col_major(b) if layout else b).to(dev)
        ad = a.to(dev)
        base = torch.zeros(M, N, dtype=torch.half, device=dev)
        host.hgemm_variant(8, layout, 1, 64, 4, ad, bb, base, swizzle=1, swizzle_stride=512)
        check(base, a, b)
        for kind, tile, bk, st in ((10, 1, 64, 2), (0, 4, 64, 2), (0, 4, 32, 4)):
            c = torch.zeros(M, N, dtype=torch.half, device=dev)
            host.hgemm_variant(kind, layout, tile, bk, st, ad, bb, c, swizzle=1, swizzle_stride=512)
            assert torch.equal(c, base), (kind, tile, bk, st)


@pytest.mark.parametrize("layout", [0, 1])
def test_one_wave_per_simd_kernel(hg, built, dev, layout):
    """hgemm_w4 (csrc/hgemm_w4.cuh, what the top rung runs at stages = 2 when K % 64 == 0 and K >= 384; 448 when K / 64 is odd): smallest legal
    K (two peeled tiles + one loop pair + two peeled tiles), odd pair counts, rectangular grids, every probe schedule incl.
    the back-to-back-read ones that exposed the zero-fill hazard; bit-identical to the ping-pong kernel (same MFMA shape
    and K order) and repeatable over launches."""
    from cuda_learn_notes_amd import host
    from cuda_learn_notes_amd.bench_utils import as_col_major
    name = ("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4" if layout
            else "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem")
    fn = getattr(hg, name)
    via_name = 0
    for (M, N, K) in ((256, 256, 384), (256, 512, 512), (768, 256, 640), (512, 768, 1152), (1024, 1024, 2048),
                      (4096, 4096, 384), (4096, 3584, 896), (256, 256, 448), (512, 256, 576), (4096, 4096, 704)):  # last three: odd K / 64
        runs_w4 = built.manifest.describe(name, (M, N, K), 2).startswith("hgemm_w4")  # the tile policy decides by M, N
        via_name += runs_w4
        a, b = seeded(170 + K, M, K), seeded(171 + K, K, N)
        bb = (as_col_major(b) if layout else b).to(dev)
        ad = a.to(dev)
        base = torch.zeros(M, N, dtype=torch.half, device=dev)
        host.hgemm_variant(8, layout, 1, 64, 4, ad, bb, base, swizzle=1, swizzle_stride=512)  # ping-pong kernel
        check(base, a, b)
        for rep in range(4 if runs_w4 else 0):
            c = torch.zeros(M, N, dtype=torch.half, device=dev)
            fn(ad, bb, c, 2, bool(rep & 1), 512)
            assert torch.equal(c, base), (M, N, K, rep)
        # odd tile counts: production schedule only; 203 / 204 = production schedule with non-temporal (what ships) / write-through C stores
        for var in ((26, 203, 204) if (K // 64) & 1 else (0, 1, 3, 4, 9, 10, 13, 20, 25, 26, 27, 28, 203, 204)):
            c = torch.zeros(M, N, dtype=torch.half, device=dev)
            host.hgemm_variant(14, layout, 1, 64, var, ad, bb, c, swizzle=1, swizzle_stride=512)
            assert torch.equal(c, base), (M, N, K, var)
    assert via_name >= 2
    # K too short for the peeled structure (< 6 tiles, or 5 = odd and < 7), or not a multiple of 64: the ping-pong kernel
    # answers, same result
    for (M, N, K) in ((4096, 4096, 320), (4096, 4096, 256), (256, 256, 352)):
        if M == 4096:
            assert built.manifest.describe(name, (M, N, K), 2).startswith("hgemm_pp"), (M, N, K)
        a, b = seeded(180 + K, M, K), seeded(181 + K, K, N)
        bb = (as_col_major(b) if layout else b).to(dev)
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        fn(a.to(dev), bb, c, 2, True, 512)
        check(c, a, b)
        with pytest.raises(RuntimeError):
            host.hgemm_variant(14, layout, 1, 64, 26, a.to(dev), bb, c, swizzle=1, swizzle_stride=512)


@pytest.mark.parametrize("layout", [0, 1])
def test_stages_3_4_5_are_the_ring_of_slots_form_of_the_one_wave_per_simd_kernel(hg, built, dev, layout):
    """hgemm_w4s (csrc/hgemm_w4s.cuh): `stages` = 3 / 4 / 5 on the 256x256 tile is the ring depth of ONE kernel template (reference
    hgemm_mma_stage.cu:2418-2428 `case 2/3/4/5`), bit-identical to stages = 2 (same MFMA shape, every accumulator sees the 32-deep
    k-steps in ascending order). Smallest legal K (2 S slots), K / 32 that leaves every remainder of the unrolled round (S = 3: rounds of
    6 bodies, S = 5: of 10), rectangular grids, block swizzle on and off; the S = 2 form through the probe hook; K too short -> another
    kernel answers, same bits."""
    from cuda_learn_notes_amd import host
    from cuda_learn_notes_amd.bench_utils import as_col_major
    name = ("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4" if layout
            else "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem")
    fn = getattr(hg, name)
    via_name = 0
    for (M, N, K) in ((256, 256, 384), (512, 256, 448), (256, 768, 512), (768, 512, 640), (256, 256, 704), (1024, 1024, 2048), (4096, 4096, 832),
                      (3584, 4096, 448), (4096, 4096, 384), (512, 512, 576), (256, 512, 896)):  # K / 32 mod 10 = 8 (576, 896): the last remainder of the S = 5 round
        a, b = seeded(370 + K, M, K), seeded(371 + K, K, N)
        bb = (as_col_major(b) if layout else b).to(dev)
        ad = a.to(dev)
        base = torch.zeros(M, N, dtype=torch.half, device=dev)
        fn(ad, bb, base, 2, True, 512)
        check(base, a, b)
        for st in (3, 4, 5):
            d = built.manifest.describe(name, (M, N, K), st)  # (the tile policy decides by M, N: small grids run a small-tile ring)
            if built.manifest.describe(name, (M, N, K), 2).startswith("hgemm_w4<256x256"):
                assert d.startswith("hgemm_w4s<256x256,ring of %d" % st) == (K // 32 >= 2 * st), (M, N, K, st, d)
                via_name += d.startswith("hgemm_w4s")
            for rep in range(2):
                c = torch.zeros(M, N, dtype=torch.half, device=dev)
                fn(ad, bb, c, st, bool(rep), 512)
                assert torch.equal(c, base), (M, N, K, st, rep)
        for S in (2, 3, 4, 5):  # the explicit hook (S = 2 exists only there)
            if K // 32 < 2 * S:
                continue
            c = torch.zeros(M, N, dtype=torch.half, device=dev)
            host.hgemm_variant(16, layout, 0, 32, S, ad, bb, c, swizzle=1, swizzle_stride=512)
            assert torch.equal(c, base), (M, N, K, S)
    assert via_name >= 6
    # the 256x256 name of the reference's WMMA stage kernel takes the same kernel at stages 3 / 4 / 5
    wn = "hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem"
    if not layout:
        a, b = seeded(390, 512, 640), seeded(391, 640, 512)
        base = torch.zeros(512, 512, dtype=torch.half, device=dev)
        getattr(hg, wn)(a.to(dev), b.to(dev), base, 2, True, 512)
        for st in (3, 4, 5):
            assert built.manifest.describe(wn, (512, 512, 640), st).startswith("hgemm_w4s<256x256,ring of %d" % st)
            c = torch.zeros(512, 512, dtype=torch.half, device=dev)
            getattr(hg, wn)(a.to(dev), b.to(dev), c, st, True, 512)
            assert torch.equal(c, base), st


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("tile,BM,BN", [(0, 192, 256), (1, 256, 192), (2, 192, 192), (3, 128, 256), (4, 256, 128), (5, 160, 160)])
def test_one_wave_per_simd_kernel_on_192_tiles(hg, built, dev, layout, tile, BM, BN):
    """The 96-row / 96-column wave tiles of hgemm_w4 (what the tile policy picks at 2304 / 3072 / 4608 / 6144): grids of
    one and several tiles, smallest and odd K-pair counts, the 384-byte-row NN image (its own bank swizzle), the
    5-rows-per-store epilogue; against the fp32 oracle and bit-identical to the 128x128 ring kernel where that tiles."""
    from cuda_learn_notes_amd import host
    from cuda_learn_notes_amd.bench_utils import as_col_major
    for (mt, nt_, K) in ((1, 1, 384), (2, 3, 640), (4, 2, 1152), (5, 5, 512), (1, 2, 448), (3, 1, 704)):
        M, N = mt * BM, nt_ * BN
        a, b = seeded(270 + K + tile, M, K), seeded(271 + K + tile, K, N)
        bb = (as_col_major(b) if layout else b).to(dev)
        ad = a.to(dev)
        first = None
        for rep in range(3):
            c = torch.zeros(M, N, dtype=torch.half, device=dev)
            host.hgemm_variant(15, layout, tile, 64, 2, ad, bb, c, swizzle=rep & 1, swizzle_stride=2 * BN)
            if first is None:
                check(c, a, b)
                first = c
            else:
                assert torch.equal(c, first), (M, N, K, rep)
        if M % 128 == 0 and N % 128 == 0:
            ring = torch.zeros(M, N, dtype=torch.half, device=dev)
            host.hgemm_variant(0, layout, 0, 64, 2, ad, bb, ring, swizzle=1, swizzle_stride=256)
            assert torch.equal(first, ring), (M, N, K)
    with pytest.raises(RuntimeError):  # not a multiple of the tile
        c = torch.zeros(256, 256, dtype=torch.half, device=dev)
        host.hgemm_variant(15, layout, 2, 64, 2, seeded(1, 256, 384).to(dev), seeded(2, 384, 256).to(dev), c, swizzle=1,
                           swizzle_stride=256)


@pytest.mark.parametrize("name,layout,BM,BN", [
    ("hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem", 0, 256, 256),
    ("hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem", 0, 256, 128),
    ("hgemm_mma_stages_block_swizzle_tn_cute", 1, 128, 256)])
def test_fixed_tile_rungs_on_the_one_wave_per_simd_kernel(hg, built, dev, name, layout, BM, BN):
    """The rungs whose name fixes the block tile (reference 256x256 / 256x128 WMMA stage kernels, the CuTe 128x256 TN
    kernel) run hgemm_w4 of THAT tile at stages = 2 when K has >= 6 (even) / >= 7 (odd) whole 64-wide tiles, the ring of the same tile otherwise;
    both answers against the fp32 oracle, and identical to each other (same MFMA shape and K order)."""
    from cuda_learn_notes_amd.bench_utils import as_col_major
    fn = getattr(hg, name)
    for (mt, nt_, K) in ((1, 1, 384), (3, 2, 640), (2, 5, 1152), (2, 1, 448), (1, 3, 832)):
        M, N = mt * BM, nt_ * BN
        assert built.manifest.describe(name, (M, N, K), 2).startswith("hgemm_w4<%dx%dx64" % (BM, BN)), (M, N, K)
        # stages 3: the ring-of-slots form of the one-wave-per-SIMD kernel on the 256x256 tile, the 8-wave ring of the tile on the others
        assert built.manifest.describe(name, (M, N, K), 3).startswith("hgemm_w4s<256x256,ring of 3" if (BM, BN) == (256, 256) else "mfma_ring<%dx%d" % (BM, BN)), (M, N, K)
        a, b = seeded(370 + K, M, K), seeded(371 + K, K, N)
        bb = (as_col_major(b) if layout else b).to(dev)
        ad = a.to(dev)
        c2 = torch.zeros(M, N, dtype=torch.half, device=dev)
        fn(ad, bb, c2, 2, True, 2 * BN)
        check(c2, a, b)
        c3 = torch.zeros(M, N, dtype=torch.half, device=dev)
        fn(ad, bb, c3, 3, False, 0)
        assert torch.equal(c2, c3), (M, N, K)
    # K outside the kernel's structure: the ring answers at stages = 2 as well
    M, N, K = BM, BN, 320
    assert built.manifest.describe(name, (M, N, K), 2).startswith("mfma_ring<"), (M, N, K)
    a, b = seeded(380, M, K), seeded(381, K, N)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    fn(a.to(dev), (as_col_major(b) if layout else b).to(dev), c, 2, False, 0)
    check(c, a, b)


@pytest.mark.parametrize("size", [2304, 2560, 2816, 3072, 3200, 4608])
def test_policy_sizes_that_run_the_192_tiles(hg, built, dev, size):
    """Through the reference names (NN and TN) at sizes where best_plan picks a 192 tile: sampled rows vs fp32."""
    from cuda_learn_notes_amd.bench_utils import as_col_major
    name = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    assert built.manifest.describe(name, (size, size, size), 2).startswith(("hgemm_w4<192x", "hgemm_w4<128x", "hgemm_w4<160x"))
    g = torch.Generator().manual_seed(size)
    a = torch.randn(size, size, generator=g).half()
    b = torch.randn(size, size, generator=g).half()
    rows = torch.arange(0, size, max(1, size // 96))[:96]
    truth = a[rows].float() @ b.float()
    ad, bd = a.to(dev), b.to(dev)
    for fn, bb in ((getattr(hg, name), bd), (getattr(hg, name + "_tn_swizzle_x4"), as_col_major(b).to(dev))):
        c = torch.zeros(size, size, dtype=torch.half, device=dev)
        fn(ad, bb, c, 2, True, 2048)
        err = (c[rows.to(dev)].cpu().float() - truth).abs()
        assert (err <= ATOL + RTOL * truth.abs()).all(), err.max().item()


def test_seeded_fuzz_over_shapes_stages_and_swizzle(hg, built, dev):
    """40 seeded random problems (M, N, K multiples of 64 up to 3072 / 3072 / 6144, stages 2-5, block swizzle on / off, stride from the
    reference policy or a random band): whatever kernel the policy picks (cln_describe names it in the failure message), sampled rows match the
    fp32 product, TN equals NN bit for bit, and every `stages` value of the same problem gives the same bits."""
    import random
    from cuda_learn_notes_amd.bench_utils import as_col_major, make_block_swizzle_stride
    rng = random.Random(20260923)
    nn_name = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    nn, tn = getattr(hg, nn_name), hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4
    kinds = set()
    for case in range(40):
        if case % 3 == 0:  # large multiples of 256: the one-wave-per-SIMD kernels and their ring-of-slots / ping-pong siblings
            M, N = 256 * rng.randint(8, 16), 256 * rng.randint(8, 16)
        else:
            M, N = 64 * rng.randint(1, 48), 64 * rng.randint(1, 48)
        K = 64 * rng.choice([1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 13, 16, 20, 26, 32, 40, 64, 96])
        stages, swz = rng.randint(2, 5), rng.random() < 0.7
        stride = make_block_swizzle_stride(N, K) if rng.random() < 0.5 else 256 * rng.randint(1, 8)
        what = built.manifest.describe(nn_name, (M, N, K), stages)
        kinds.add(what.split("<")[0])
        a, b = seeded(1000 + case, M, K), seeded(2000 + case, K, N)
        ad, bd = a.to(dev), b.to(dev)
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        nn(ad, bd, c, stages, swz, stride)
        rows = sorted({0, M - 1, M // 2, rng.randrange(M), rng.randrange(M)})
        truth = a[rows].float() @ b.float()
        err = (c[rows].cpu().float() - truth).abs()
        assert (err <= ATOL + RTOL * truth.abs()).all(), (case, (M, N, K), stages, what, err.max().item())
        ct = torch.zeros(M, N, dtype=torch.half, device=dev)
        tn(ad, as_col_major(b).to(dev), ct, stages, swz, stride)
        assert torch.equal(ct, c), (case, (M, N, K), stages, what)
        other = 2 + (stages - 1) % 4  # another stage count of the same problem
        co = torch.zeros(M, N, dtype=torch.half, device=dev)
        nn(ad, bd, co, other, not swz, stride)
        assert torch.equal(co, c), (case, (M, N, K), stages, other, what, built.manifest.describe(nn_name, (M, N, K), other))
    assert len(kinds) >= 3, kinds  # the sample reached several kernel families


SPLIT_K_SHAPES = [(1024, 1024, 16384), (128, 8192, 8192), (256, 4096, 4096), (768, 768, 12288), (640, 5120, 5120), (1536, 1536, 8192),
                  (2048, 2048, 16384), (256, 256, 16384), (1024, 1024, 4160), (384, 768, 4480), (2048, 2048, 8960), (1536, 2048, 8192),
                  (192, 256, 16384), (960, 960, 4480), (1344, 1344, 4480)]


@pytest.mark.parametrize("M,N,K", SPLIT_K_SHAPES)
def test_split_k_full_matrix(hg, built, dev, M, N, K):
    """Few output tiles, long K (csrc/hgemm_splitk.cuh: K split over S workgroups per tile of the one-wave-per-SIMD kernel, fp32 partials in
    register layout, one reduce launch): the FULL matrix against the fp32 product at every tile shape the plan uses (256x256, 192x256, 192x192,
    128x256, 160x160), even and odd numbers of K tiles per split, S from 2 to 32; TN equals NN bit for bit; `stages` is ignored."""
    from cuda_learn_notes_amd.bench_utils import as_col_major
    name = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    what = built.manifest.describe(name, (M, N, K), 2)
    assert "split-K x " in what and ("hgemm_splitk_reduce" in what or "in-kernel fix-up" in what), what
    a, b = seeded(M + K, M, K), seeded(N + K, K, N)
    ad, bd = a.to(dev), b.to(dev)
    c = torch.full((M, N), float("nan"), dtype=torch.half, device=dev)
    getattr(hg, name)(ad, bd, c, 2, True, 2048)
    check(c, a, b)
    ct = torch.full((M, N), float("nan"), dtype=torch.half, device=dev)
    hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4(ad, as_col_major(b).to(dev), ct, 4, False, 0)
    assert torch.equal(ct, c), what


def test_split_k_workspace_grows_and_is_per_stream(hg, built, dev):
    """One fp32 workspace per stream -- since round 6 a tensor of torch's caching allocator that host.py registers with the library (the library
    itself allocates nothing), grown on demand: a small problem, then a larger one, then the small one again on the same stream, and two problems
    interleaved on two streams, all give the single-stream results; the library holds no memory of its own at any point."""
    from cuda_learn_notes_amd import host
    name = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    fn = getattr(hg, name)
    host.release_workspaces()
    probs = []
    for (M, N, K) in ((256, 256, 8192), (1024, 1024, 16384), (256, 512, 4096)):
        a, b = seeded(M + 7, M, K).to(dev), seeded(N + 9, K, N).to(dev)
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        fn(a, b, c, 2, False, 0)
        probs.append((a, b, c))
    torch.cuda.synchronize()
    for (a, b, c) in probs:
        check(c, a.cpu(), b.cpu())
    (key, size), = host.hgemm_workspace_tensors().items()
    assert size >= host.hgemm_workspace_bytes(1024, 1024, 16384) and host.hgemm_workspace_held() == 0
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rnd in range(3):
        for st, (a, b, c) in ((s1, probs[0]), (s2, probs[1]), (s1, probs[2]), (s2, probs[0])):
            with torch.cuda.stream(st):
                o = torch.zeros_like(c)  # (filled on the stream that runs the launch: a fill on another stream races with it)
                fn(a, b, o, 2, False, 0)
            outs.append((o, c))
    torch.cuda.synchronize()
    for o, c in outs:
        assert torch.equal(o, c)
    assert len(host.hgemm_workspace_tensors()) == 3 and host.hgemm_workspace_held() == 0


@pytest.mark.parametrize("M,N,K,eager_form", [(512, 512, 8192, "hgemm_splitk_reduce"), (2048, 2048, 8192, "in-kernel fix-up"), (640, 5120, 5120, "in-kernel fix-up")])
def test_split_k_under_stream_capture(hg, built, dev, M, N, K, eager_form):
    """A captured launch replays: the workspace the stream already has is used inside a graph (always as partial + reduce launch: no arrival
    tickets in a graph, ADVICE r5 -- the same bits as the eager one-launch form); a stream that has none yet (nothing is allocated while
    capturing) takes the single-pass plan -- same result within the parity tolerance either way."""
    from cuda_learn_notes_amd import host
    name = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    fn = getattr(hg, name)
    assert eager_form in built.manifest.describe(name, (M, N, K), 2)  # eager: more than 2 splits -> two launches, 2 splits -> one launch with tickets
    a, b = seeded(3, M, K), seeded(4, K, N)
    ad, bd = a.to(dev), b.to(dev)
    for warm in (True, False):
        st = torch.cuda.Stream()
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        eager = None
        with torch.cuda.stream(st):
            if warm:
                fn(ad, bd, c, 2, False, 0)  # the stream's workspace now exists
                st.synchronize()
                eager = c.clone()
                c.zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                fn(ad, bd, c, 2, False, 0)
            g.replay()
            g.replay()
        torch.cuda.synchronize()
        check(c, a, b)
        if warm:
            assert torch.equal(c, eager)
            with torch.cuda.stream(st):  # the eager one-launch form still works on the stream the graph came from (its tickets were never in a graph)
                for _ in range(3):
                    c.zero_()
                    g.replay()
                    fn(ad, bd, c, 2, False, 0)
            torch.cuda.synchronize()
            assert torch.equal(c, eager)
    assert host.hgemm_workspace_held() == 0


@pytest.mark.parametrize("owner", ["torch", "library"])
def test_workspace_of_a_captured_graph_is_never_evicted(hg, built, dev, owner):
    """A graph captured on a stream that has a workspace holds that workspace's ADDRESS. Twelve other streams running split-K afterwards would push
    it out of the 8-entry LRU: a workspace a capture has used is pinned, so the replayed graph still writes into live memory and still returns the
    right product. owner = torch: the tensors host.py keeps per stream (the default: the library holds nothing). owner = library: the C caller's
    opt-in, cln_hgemm_library_workspace(1), driven through the raw C-ABI (csrc/hgemm.hip SplitKWs::pinned); cln_release_workspaces() frees it."""
    from cuda_learn_notes_amd import _loader, host
    name = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    M, N, K = 512, 512, 8192
    assert host.hgemm_workspace_bytes(M, N, K) > 0
    a, b = seeded(41, M, K), seeded(42, K, N)
    ad, bd = a.to(dev), b.to(dev)
    host.release_workspaces()
    if owner == "torch":
        hfn = getattr(hg, name)

        def fn(c):
            hfn(ad, bd, c, 2, False, 0)
    else:
        raw = _loader.symbol(name)
        assert host.hgemm_library_workspace(True) is False

        def fn(c):
            assert raw(ad.data_ptr(), bd.data_ptr(), c.data_ptr(), M, N, K, 2, 0, 0, torch.cuda.current_stream().cuda_stream) == 0
    try:
        st = torch.cuda.Stream()
        c = torch.zeros(M, N, dtype=torch.half, device=dev)
        with torch.cuda.stream(st):
            fn(c)  # the stream's workspace now exists
            st.synchronize()
            ref = c.clone()
            one = host.hgemm_workspace_held()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                fn(c)
        check(ref, a, b)
        assert (one == 0) if owner == "torch" else (one == 4096 + (16 << 20)), one
        others = [torch.cuda.Stream() for _ in range(12)]
        scratch = torch.zeros(M, N, dtype=torch.half, device=dev)
        for o in others:
            with torch.cuda.stream(o):
                fn(scratch)
            o.synchronize()
        if owner == "torch":
            assert host.hgemm_workspace_held() == 0 and len(host.hgemm_workspace_tensors()) == 9  # the 8 evictable ones + the pinned one
            assert (dev.index or 0, st.cuda_stream) in host.hgemm_workspace_tensors()
        else:
            assert host.hgemm_workspace_held() == 9 * one and not host.hgemm_workspace_tensors()
        with torch.cuda.stream(st):
            for _ in range(3):
                c.zero_()
                g.replay()
        torch.cuda.synchronize()
        assert torch.equal(c, ref)
        del g
        freed = host.release_workspaces()
        assert (freed > 0) == (owner == "library") or owner == "torch"  # (torch: only scratch slabs, if any, are the library's to free)
        assert host.hgemm_workspace_held() == 0 and not host.hgemm_workspace_tensors()
    finally:
        host.hgemm_library_workspace(False)


@pytest.mark.parametrize("M,N,K", [(4352, 4352, 4352), (5888, 5888, 1792), (4096, 4352, 2048), (4864, 4864, 4864)])
def test_tail_split_matches_the_fp32_product(hg, built, dev, M, N, K):
    """A count of 256 x 256 tiles just past whole rounds of 256 (csrc/hgemm.hip tail_plan): the rows that fill whole rounds run the single-pass
    kernel, the last tile rows run split-K; sampled rows from both regions and the seam against the fp32 product, TN equals NN bit for bit,
    every `stages` value gives the same bits (the plan ignores it)."""
    import re
    from cuda_learn_notes_amd.bench_utils import as_col_major, make_block_swizzle_stride
    name = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    what = built.manifest.describe(name, (M, N, K), 2)
    assert "tail split" in what, what
    m_split = int(re.search(r"on rows \[0, (\d+)\)", what).group(1))
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    rows = torch.unique(torch.tensor([0, 255, m_split // 2, m_split - 1, m_split, m_split + 1, m_split + 255, (m_split + M) // 2, M - 256, M - 1]))
    truth = a[rows].float() @ b.float()
    stride = make_block_swizzle_stride(N, K)
    c = torch.full((M, N), float("nan"), dtype=torch.half, device=dev)
    getattr(hg, name)(a, b, c, 2, True, stride)
    assert not torch.isnan(c).any()
    err = (c[rows].float() - truth).abs()
    assert (err <= ATOL + RTOL * truth.abs()).all(), (what, err.max().item())
    ct = torch.zeros(M, N, dtype=torch.half, device=dev)
    hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4(a, as_col_major(b), ct, 3, False, 0)
    assert torch.equal(ct, c), what


# ------------------------------------------------------------------ split-K workspace entry points (round 5)
def _hip_runtime():
    """torch's own HIP runtime (the one libcln_amd.so is bound to: _loader imports torch first), for raw stream handles."""
    import ctypes
    import glob
    import os
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*"))
    return ctypes.CDLL(cands[0]) if cands else ctypes.CDLL("libamdhip64.so")


def test_workspace_entry_points_caller_owned_and_release(hg, built, dev):
    """include/cln_amd.h workspace block: cln_hgemm_workspace_bytes says what a shape needs (0 for a single-pass shape); by default the LIBRARY
    allocates nothing (round 6) -- host.py registers one torch tensor per stream; a region the caller registers itself gives the same bits, a region
    that is too small makes the shape run single-pass (still inside the parity tolerance); through the raw C-ABI a stream without a region runs
    single-pass until cln_hgemm_library_workspace(1) opts in to library-owned buffers (4 KiB + a power of two: ADVICE r5), which
    cln_release_workspaces frees."""
    from cuda_learn_notes_amd import _loader, host
    name = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    fn = getattr(hg, name)
    raw = _loader.symbol(name)
    assert host.hgemm_workspace_bytes(4096, 4096, 4096) == 0  # 256 tiles, one round: single pass
    M, N, K = 1024, 1024, 16384
    need = host.hgemm_workspace_bytes(M, N, K)
    S = int(re.search(r"split-K x (\d+)", built.manifest.describe(name, (M, N, K), 2)).group(1))
    assert need == 4096 + S * M * N * 4
    assert host.hgemm_workspace_bytes(4352, 4352, 4352) > 0  # tail split
    a, b = seeded(21, M, K), seeded(22, K, N)
    ad, bd = a.to(dev), b.to(dev)
    st = torch.cuda.Stream()
    key = (dev.index or 0, st.cuda_stream)
    with torch.cuda.stream(st):
        host.release_workspaces()
        assert host.hgemm_workspace_held() == 0 and not host.hgemm_workspace_tensors()
        c_auto = torch.zeros(M, N, dtype=torch.half, device=dev)
        fn(ad, bd, c_auto, 2, False, 0)
        st.synchronize()
        assert host.hgemm_workspace_held() == 0  # nothing hidden
        assert need <= host.hgemm_workspace_tensors()[key] <= max(16 << 20, need + (2 << 20))
        buf = torch.empty(need, dtype=torch.uint8, device=dev)
        buf.fill_(0xAB)  # garbage: the library zeroes the ticket header itself
        host.hgemm_set_workspace(buf)
        assert host.hgemm_workspace_tensors()[key] == need
        c_usr = torch.zeros(M, N, dtype=torch.half, device=dev)
        for _ in range(3):  # self-resetting tickets: repeated launches on the same region
            c_usr.zero_()
            fn(ad, bd, c_usr, 2, False, 0)
        st.synchronize()
        assert torch.equal(c_usr, c_auto)
        small = torch.empty(8192, dtype=torch.uint8, device=dev)
        host.hgemm_set_workspace(small)  # too small for this shape, and the caller's: never replaced -> single-pass plan
        c_sp = torch.zeros(M, N, dtype=torch.half, device=dev)
        fn(ad, bd, c_sp, 2, False, 0)
        st.synchronize()
        assert host.hgemm_workspace_tensors()[key] == 8192 and host.hgemm_workspace_held() == 0
        host.hgemm_set_workspace(None)
        assert key not in host.hgemm_workspace_tensors()
        # the raw C-ABI on a stream nobody gave a region: single-pass (the bits of c_sp), no allocation
        c_raw = torch.zeros(M, N, dtype=torch.half, device=dev)
        assert raw(ad.data_ptr(), bd.data_ptr(), c_raw.data_ptr(), M, N, K, 2, 0, 0, st.cuda_stream) == 0
        st.synchronize()
        assert torch.equal(c_raw, c_sp) and host.hgemm_workspace_held() == 0
        # ... until the C caller opts in
        assert host.hgemm_library_workspace(True) is False
        try:
            c_lib = torch.zeros(M, N, dtype=torch.half, device=dev)
            assert raw(ad.data_ptr(), bd.data_ptr(), c_lib.data_ptr(), M, N, K, 2, 0, 0, st.cuda_stream) == 0
            st.synchronize()
            p2 = 16 << 20
            while p2 < need - 4096:
                p2 <<= 1
            assert torch.equal(c_lib, c_auto) and host.hgemm_workspace_held() == 4096 + p2
            assert host.release_workspaces() >= 4096 + p2 and host.hgemm_workspace_held() == 0
        finally:
            assert host.hgemm_library_workspace(False) is True
        fn(ad, bd, c_usr, 2, False, 0)  # host.py's own tensor again
        st.synchronize()
        assert torch.equal(c_usr, c_auto) and host.hgemm_workspace_held() == 0
    check(c_auto, a, b)
    check(c_sp, a, b)
    host.release_workspaces()


def test_workspace_does_not_leak_over_many_streams(hg, built, dev):
    """a process that cycles streams must not pin a workspace per stream for ever. (1) Library-owned buffers (the C caller's
    opt-in): 300 raw HIP streams, each created, used for one split-K launch and destroyed -- the library never holds more than 8 workspaces (least
    recently used freed once its last launch has completed: the completion event outlives the stream), every result is right. (2) host.py's torch
    tensors: 40 torch streams, at most 8 tensors registered at a time."""
    import ctypes
    from cuda_learn_notes_amd import _loader, host
    hip = _hip_runtime()
    hip.hipStreamCreate.argtypes, hip.hipStreamCreate.restype = [ctypes.POINTER(ctypes.c_void_p)], ctypes.c_int
    hip.hipStreamDestroy.argtypes, hip.hipStreamDestroy.restype = [ctypes.c_void_p], ctypes.c_int
    hip.hipStreamSynchronize.argtypes, hip.hipStreamSynchronize.restype = [ctypes.c_void_p], ctypes.c_int
    raw = _loader.symbol("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem")
    fn = getattr(hg, "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem")
    M, N, K = 256, 256, 8192
    assert host.hgemm_workspace_bytes(M, N, K) > 0
    a, b = seeded(31, M, K), seeded(32, K, N)
    ad, bd = a.to(dev), b.to(dev)
    ref = torch.zeros(M, N, dtype=torch.half, device=dev)
    fn(ad, bd, ref, 2, False, 0)
    torch.cuda.synchronize()
    check(ref, a, b)
    host.release_workspaces()
    one = 4096 + (16 << 20)
    peak = 0
    outs = [torch.zeros(M, N, dtype=torch.half, device=dev) for _ in range(4)]
    assert host.hgemm_library_workspace(True) is False
    try:
        for i in range(300):
            s = ctypes.c_void_p()
            assert hip.hipStreamCreate(ctypes.byref(s)) == 0
            c = outs[i % 4]
            assert raw(ad.data_ptr(), bd.data_ptr(), c.data_ptr(), M, N, K, 2, 0, 0, s) == 0
            if i % 4 == 3:
                assert hip.hipStreamSynchronize(s) == 0
                assert torch.equal(c, ref), i
            peak = max(peak, host.hgemm_workspace_held())
            if i % 2 == 0:  # half of the streams are destroyed while their launch may still be queued
                assert hip.hipStreamDestroy(s) == 0
        torch.cuda.synchronize()
        assert 0 < peak <= 8 * one, peak
        assert host.release_workspaces() <= 8 * one + (64 * 16384) * 2 and host.hgemm_workspace_held() == 0
    finally:
        host.hgemm_library_workspace(False)
    streams = [torch.cuda.Stream() for _ in range(40)]
    res = []
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            o = torch.zeros(M, N, dtype=torch.half, device=dev)
            fn(ad, bd, o, 2, False, 0)
            res.append(o)
        assert len(host.hgemm_workspace_tensors()) <= 8 and host.hgemm_workspace_held() == 0
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in res)
    host.release_workspaces()


def test_two_host_threads_on_one_stream_get_their_own_results(hg, built, dev):
    """ctypes drops the GIL around the C call, so two threads can be inside hgemm on the same stream; the workspace lock
    covers each call's whole launch sequence, so the partial / reduce launches of two calls never interleave."""
    import threading
    name = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    fn = getattr(hg, name)
    probs = []
    for seed, (M, N, K) in enumerate(((512, 512, 8192), (256, 256, 16384))):  # both split-K, S above and below the one-launch limit
        a, b = seeded(40 + seed, M, K).to(dev), seeded(50 + seed, K, N).to(dev)
        ref = torch.zeros(M, N, dtype=torch.half, device=dev)
        fn(a, b, ref, 2, False, 0)
        probs.append((a, b, ref))
    torch.cuda.synchronize()
    outs = [[], []]
    st = torch.cuda.Stream()

    def work(i):
        a, b, ref = probs[i]
        with torch.cuda.stream(st):
            for _ in range(40):
                o = torch.zeros_like(ref)
                fn(a, b, o, 2, False, 0)
                outs[i].append(o)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    for i in range(2):
        assert len(outs[i]) == 40 and all(torch.equal(o, probs[i][2]) for o in outs[i])
