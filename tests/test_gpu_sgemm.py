"""GPU parity: SGEMM ladder through the C-ABI vs the fp64 oracle. Every rung accumulates in
exact fp32 (the TF32-named rungs run on v_mfma_f32_32x32x2_f32), so the tolerance is the fp32 summation error:
|err| <= 4e-6 * sqrt(K) * |a|_rms*|b|_rms-ish -> we assert 2e-5 * sqrt(K) on N(0,1) operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(built, dev):
    return built.load("sgemm", "sgemm_vendor")


def seeded(seed, *shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


S3 = ["sgemm_naive_f32", "sgemm_sliced_k_f32", "sgemm_t_8x8_sliced_k_f32x4", "sgemm_t_8x8_sliced_k_f32x4_bcf",
      "sgemm_t_8x8_sliced_k_f32x4_bcf_offset", "sgemm_t_8x8_sliced_k_f32x4_bcf_dbuf",
      "sgemm_t_8x8_sliced_k_f32x4_bcf_dbuf_offset", "sgemm_t_8x4_sliced_k16_f32x4_bcf_dbuf",
      "sgemm_t_8x4_sliced_k16_f32x4_bcf_dbuf_async", "sgemm_t_8x8_sliced_k16_f32x4_bcf_dbuf",
      "sgemm_t_8x8_sliced_k16_f32x4_bcf_dbuf_async", "sgemm_t_8x16_sliced_k16_f32x4_bcf_dbuf",
      "sgemm_t_8x16_sliced_k16_f32x4_bcf_dbuf_async", "sgemm_cublas", "sgemm_cublas_tf32"]


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 320), (1024, 512, 1024)])
def test_sgemm_all_rungs(lib, dev, oracle, M, N, K):
    a, b = seeded(M + K, M, K), seeded(N + K, K, N)
    ref = oracle.sgemm(a, b)
    tol = 2e-5 * K ** 0.5
    ad, bd = a.to(dev), b.to(dev)
    for name in S3:
        c = torch.zeros(M, N, device=dev)
        getattr(lib, name)(ad, bd, c)
        assert (c.cpu().double() - ref).abs().max().item() <= tol, name
    for name in ("sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages", "sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages_dsmem"):
        first = None
        for stages in (2, 3):
            for swz in (False, True):
                c = torch.zeros(M, N, device=dev)
                getattr(lib, name)(ad, bd, c, stages, swz, 512)
                assert (c.cpu().double() - ref).abs().max().item() <= tol, (name, stages, swz)
                if first is None:
                    first = c
                else:
                    assert torch.equal(c, first)  # schedule knobs never change the k-ordered fp32 result


@pytest.mark.parametrize("M,N,K", [(64, 128, 16), (192, 384, 48), (3072, 3072, 32), (4096, 4096, 64), (8192, 8192, 48), (16384, 4096, 16)])
def test_sgemm_matrix_core_tile_forms(lib, dev, oracle, M, N, K):
    """The LDS-DMA f32-MFMA kernel (csrc/sgemm_dma.cuh) on shapes that the planner sends to each of its tile forms -- 64x128 (few tiles, or
    3072^2: 576 tiles of 128x128 would be 2.25 per CU), 128x128 (4096^2), 256x128 (>= 8192^2) -- with K of one, two, three and four 16-deep
    stages (prologue only / tail branches / one trip of the three-slot loop): fp64 oracle, and the same bits from every knob value."""
    a, b = seeded(3 * M + K, M, K), seeded(5 * N + K, K, N)
    ref = oracle.sgemm(a, b)
    tol = 2e-5 * K ** 0.5
    ad, bd = a.to(dev), b.to(dev)
    first = None
    for stages, swz in ((2, False), (3, True)):
        c = torch.full((M, N), float("nan"), device=dev)
        lib.sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages(ad, bd, c, stages, swz, 256)
        assert (c.cpu().double() - ref).abs().max().item() <= tol, (stages, swz)
        first = c if first is None else first
        assert torch.equal(c, first)


def test_sgemm_matrix_core_result_does_not_depend_on_the_tile_form(lib, dev):
    """Every tile form adds the k products of an output element in the same order, so a sub-block of a large product (256x128 tiles) equals the
    small product of the same rows and columns (64x128 / 128x128 tiles) bit for bit."""
    K = 208
    a, b = seeded(11, 8192, K).to(dev), seeded(12, K, 8192).to(dev)
    c = torch.zeros(8192, 8192, device=dev)
    lib.sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages(a, b, c, 2, True, 256)
    for (r0, rows, c0, cols) in ((0, 64, 0, 128), (4096, 512, 2048, 1024), (1024, 4096, 2048, 4096)):
        sub = torch.zeros(rows, cols, device=dev)
        lib.sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages(a[r0:r0 + rows].contiguous(), b[:, c0:c0 + cols].contiguous(), sub, 3, False, 256)
        assert torch.equal(sub, c[r0:r0 + rows, c0:c0 + cols]), (rows, cols)


def test_sgemm_identity_asymmetric_exact(lib, dev):
    n = 256
    a = torch.eye(n)
    b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 2039 - 1000) / 8
    for name in ("sgemm_t_8x8_sliced_k16_f32x4_bcf_dbuf_async", "sgemm_sliced_k_f32"):
        c = torch.zeros(n, n, device=dev)
        getattr(lib, name)(a.to(dev), b.to(dev), c)
        assert torch.equal(c.cpu(), b), name
    c = torch.zeros(n, n, device=dev)
    lib.sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages(a.to(dev), b.to(dev), c, 2, False, 1)
    assert torch.equal(c.cpu(), b)


def test_sgemm_unsupported_shape(lib, dev):
    a, b, c = torch.zeros(100, 64, device=dev), torch.zeros(64, 128, device=dev), torch.zeros(100, 128, device=dev)
    with pytest.raises(RuntimeError, match="multiples of the block tile"):
        lib.sgemm_t_8x8_sliced_k_f32x4(a, b, c)
    with pytest.raises(RuntimeError, match="values must be torch::kFloat32"):
        lib.sgemm_naive_f32(a.half(), b.half(), c.half())


@pytest.mark.parametrize("M,N,K", [(1024, 1024, 1024), (512, 2048, 544), (64, 128, 528), (1024, 512, 4096)])
def test_sgemm_matrix_core_k_split_for_few_tiles(lib, dev, oracle, M, N, K):
    """At most 128 tiles of 64x128 and K >= 512: two workgroups per tile, each half of the stages (an odd stage count: 17 + 16), their products added
    onto a zeroed C by one fp32 atomic each -- fp64 oracle, the output buffer's previous content must not matter, and 20 launches give the same
    bits (two commutative additions per element)."""
    a, b = seeded(7 * M + K, M, K), seeded(9 * N + K, K, N)
    ref = oracle.sgemm(a, b)
    tol = 2e-5 * K ** 0.5
    ad, bd = a.to(dev), b.to(dev)
    first = None
    for r in range(20):
        c = torch.full((M, N), float("nan") if r % 2 else 123.0, device=dev)
        lib.sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages(ad, bd, c, 2 + (r & 1), bool(r & 2), 256)
        if first is None:
            assert (c.cpu().double() - ref).abs().max().item() <= tol
            first = c
        assert torch.equal(c, first), r


def test_sgemm_k_split_under_stream_capture(lib, dev):
    """The split form is a memset node + a kernel node: a captured launch replays into a buffer that something else overwrote in between."""
    M = N = K = 1024
    a, b = seeded(1, M, K).to(dev), seeded(2, K, N).to(dev)
    c = torch.zeros(M, N, device=dev)
    lib.sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages(a, b, c, 2, True, 256)
    want = c.clone()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        lib.sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages(a, b, c, 2, True, 256)  # warm-up on the capture stream
        torch.cuda.current_stream().synchronize()
        with torch.cuda.graph(g, stream=s):
            lib.sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages(a, b, c, 2, True, 256)
    for _ in range(3):
        c.fill_(5.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(c, want)
