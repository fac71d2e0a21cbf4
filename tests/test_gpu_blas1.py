"""GPU parity: dot-product / sgemv / hgemv / mat-transpose through the C-ABI vs the oracle.
Transpose is BIT-EXACT (the reference checks out.T.equal(x), mat_transpose.py:60)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(built, dev):
    return built.load("dot_product", "sgemv", "hgemv", "mat_transpose")


@pytest.mark.parametrize("shape", [(1024, 1024), (4096, 2048), (3, 1000), (1, 7)])
def test_dot_product(lib, dev, oracle, shape):
    g = torch.Generator().manual_seed(shape[0] + shape[1])
    a, b = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g)
    n = a.numel()
    for name in ("dot_prod_f32_f32", "dot_prod_f32x4_f32"):
        if "x4" in name and n % 4:
            continue
        ref = oracle.dot_prod(a, b)
        got = getattr(lib, name)(a.to(dev), b.to(dev)).item()
        assert abs(got - ref) <= 1e-4 * (n ** 0.5) + 1e-5 * abs(ref), (name, got, ref)  # fp32 sum of n terms ~ N(0,1)
    ah, bh = a.half(), b.half()
    ref = oracle.dot_prod(ah, bh)
    for name in ("dot_prod_f16_f32", "dot_prod_f16x2_f32", "dot_prod_f16x8_pack_f32"):
        if (("x2" in name and n % 2) or ("x8" in name and n % 8)):
            continue
        got = getattr(lib, name)(ah.to(dev), bh.to(dev)).item()
        assert abs(got - ref) <= 1e-4 * (n ** 0.5) + 1e-5 * abs(ref), (name, got, ref)


# (the last three: the rows-per-wave kernels of the f16 rungs -- four rows per wave from 16384 rows on, two below; row counts that leave a wave's last
# rows empty, K with a remainder behind the unrolled pieces)
@pytest.mark.parametrize("M,K,names", [(1024, 128, ("k32", "k128")), (1000, 32, ("k32",)), (4096, 4096, ("k32", "k128")),
                                       (1024, 16, ("k16",)), (37, 16, ("k16",)), (20001, 1280, ("k32", "k128")), (5001, 768, ("k32", "k128")),
                                       (4099, 2048, ("k32", "k128"))])
def test_gemv(lib, dev, oracle, M, K, names):
    g = torch.Generator().manual_seed(M + K)
    a, x = torch.randn(M, K, generator=g), torch.randn(K, 1, generator=g)
    suffix = {"k32": ("f32", "f16"), "k128": ("f32x4", "f16x4"), "k16": ("f32", "f16")}
    for kk in names:
        ref = oracle.gemv(a, x)
        y = torch.zeros(M, 1, device=dev)
        getattr(lib, "sgemv_%s_%s" % (kk, suffix[kk][0]))(a.to(dev), x.to(dev), y)
        assert torch.allclose(y.cpu().double(), ref, rtol=1e-5, atol=1e-4 * K ** 0.5)
        ah, xh = a.half(), x.half()
        refh = oracle.gemv(ah, xh)
        yh = torch.zeros(M, 1, dtype=torch.half, device=dev)
        getattr(lib, "hgemv_%s_%s" % (kk, suffix[kk][1]))(ah.to(dev), xh.to(dev), yh)
        assert torch.allclose(yh.cpu().double(), refh, rtol=2e-3, atol=2e-3 * K ** 0.5)  # one fp16 rounding of the sum


def test_gemv_k_constraints(lib, dev):
    a, x, y = torch.zeros(8, 48, device=dev), torch.zeros(48, 1, device=dev), torch.zeros(8, 1, device=dev)
    with pytest.raises(RuntimeError, match="K must be multiple of 32"):
        lib.sgemv_k32_f32(a, x, y)
    with pytest.raises(RuntimeError, match="K must be 16"):
        lib.sgemv_k16_f32(a, x, y)


TR_ALL = ["mat_transpose_f32_col2row", "mat_transpose_f32x4_col2row", "mat_transpose_f32_row2col",
          "mat_transpose_f32x4_row2col", "mat_transpose_f32_col2row2d", "mat_transpose_f32x4_col2row2d",
          "mat_transpose_f32_row2col2d", "mat_transpose_f32x4_row2col2d", "mat_transpose_f32_diagonal2d",
          "mat_transpose_f32x4_shared_col2row2d", "mat_transpose_f32x4_shared_row2col2d",
          "mat_transpose_f32x4_shared_bcf_col2row2d", "mat_transpose_f32x4_shared_bcf_row2col2d"]


@pytest.mark.parametrize("M,N", [(1024, 1024), (2048, 4096), (4096, 1024), (64, 192), (1024, 2048)])
def test_transpose_bit_exact(lib, dev, oracle, M, N):
    x = (torch.arange(M * N, dtype=torch.float32).reshape(M, N) * 0.25 - 1000.0)  # asymmetric, exact in fp32
    ref = oracle.mat_transpose(x)
    xd = x.to(dev)
    for name in TR_ALL:
        y = torch.full((N, M), -1.0, device=dev)
        getattr(lib, name)(xd, y)
        assert torch.equal(y.cpu(), ref), name
        assert y.T.equal(xd)  # the reference script's own check (mat_transpose.py:60)


def test_transpose_ragged_shapes(lib, dev, oracle):
    x = torch.randn(37, 53)
    ref = oracle.mat_transpose(x)
    for name in ("mat_transpose_f32_col2row", "mat_transpose_f32_row2col", "mat_transpose_f32_diagonal2d"):
        y = torch.zeros(53, 37, device=dev)
        getattr(lib, name)(x.to(dev), y)
        assert torch.equal(y.cpu(), ref), name
    with pytest.raises(RuntimeError, match="multiples of"):
        lib.mat_transpose_f32x4_shared_col2row2d(x.to(dev), torch.zeros(53, 37, device=dev))
    # the f32x4 *2d rungs: 4 x 4 register blocks when both extents divide by 32 (round 6), the 1-D f32x4 rung of the same name otherwise
    for (M, N) in ((36, 100), (32, 96), (96, 32), (160, 224)):
        x = torch.randn(M, N)
        for name in ("mat_transpose_f32x4_col2row2d", "mat_transpose_f32x4_row2col2d"):
            y = torch.full((N, M), -7.0, device=dev)
            getattr(lib, name)(x.to(dev), y)
            assert torch.equal(y.cpu(), oracle.mat_transpose(x)), (name, M, N)
