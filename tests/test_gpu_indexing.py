"""GPU parity, BIT-EXACT: histogram and embedding through the C-ABI vs the oracle (SURVEY 8(f) rank 1;
north_star: "bit-exact for histogram/indexing")."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(built, dev):
    return built.load("histogram", "embedding")


def test_histogram_reference_readme_vector(lib, dev):
    a = torch.tensor(list(range(10)) * 1000, dtype=torch.int32, device=dev)  # reference histogram.py:22
    for fn in (lib.histogram_i32, lib.histogram_i32x4):
        h = fn(a)
        assert h.dtype == torch.int32 and h.cpu().tolist() == [1000] * 10  # README.md:24-44


@pytest.mark.parametrize("n,nbins", [(1, 1), (3, 7), (1000, 10), (4099, 257), (1 << 20, 1024), (1 << 22, 8192),
                                     (1 << 20, 8193), (1 << 21, 100000), (777777, 50000)])
def test_histogram_bit_exact(lib, dev, oracle, n, nbins):
    g = torch.Generator().manual_seed(n + nbins)
    a = torch.randint(0, nbins, (n,), generator=g, dtype=torch.int32)
    a[0] = nbins - 1  # pin max(a) so the binding sizes y like the oracle does
    ref = oracle.histogram(a)
    ad = a.to(dev)
    for fn in (lib.histogram_i32, lib.histogram_i32x4):
        for _ in range(2):  # atomics: repeat, result must not depend on scheduling
            h = fn(ad)
            assert h.shape == ref.shape and torch.equal(h.cpu(), ref), fn.__name__


def test_histogram_skewed_and_checksum(lib, dev, oracle):
    """Heavy collisions (all elements in 3 bins) and the size-independent property sum(h) == n at 64 Mi elements."""
    n = 1 << 26
    a = (torch.arange(n, dtype=torch.int32, device=dev) % 3) * 500
    h = lib.histogram_i32x4(a)
    assert int(h.sum().item()) == n and h.shape[0] == 1001
    assert h[0].item() == (n + 2) // 3 and h[500].item() == (n + 1) // 3 and h[1000].item() == n // 3
    assert int((h != 0).sum().item()) == 3


def test_histogram_dtype_error(lib, dev):
    with pytest.raises(RuntimeError, match="values must be torch::kInt32"):
        lib.histogram_i32(torch.zeros(8, device=dev))


@pytest.mark.parametrize("M,N,K", [(1024, 2048, 512), (4096, 4096, 1024), (7, 1, 8), (300, 1000, 24), (50000, 333, 4096)])
def test_embedding_bit_exact(lib, dev, oracle, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    idx = torch.randint(0, M, (N,), generator=g, dtype=torch.int32)
    idx[0], idx[-1] = M - 1, 0
    w = torch.randn(M, K, generator=g)
    for dt, names in ((torch.float32, ("embedding_f32", "embedding_f32x4", "embedding_f32x4_pack")),
                      (torch.float16, ("embedding_f16", "embedding_f16x8", "embedding_f16x8_pack"))):
        wd = w.to(dt)
        ref = oracle.embedding(idx, wd)
        for name in names:
            o = torch.full((N, K), 7.0, dtype=dt, device=dev)
            getattr(lib, name)(idx.to(dev), wd.to(dev), o)
            assert torch.equal(o.cpu(), ref), name


def test_embedding_pack_width_and_oob(lib, dev):
    w = torch.randn(16, 12, device=dev)
    idx = torch.zeros(4, dtype=torch.int32, device=dev)
    o = torch.zeros(4, 12, device=dev)
    lib.embedding_f32x4(idx, w, o)  # 12 % 4 == 0
    wh, oh = w.half(), o.half()
    with pytest.raises(RuntimeError, match="multiple of the pack width"):
        lib.embedding_f16x8(idx, wh, oh)  # 12 % 8 != 0
    idx2 = torch.tensor([1, 99, -3, 2], dtype=torch.int32, device=dev)  # out-of-range rows are zero-filled
    o.fill_(5.0)
    lib.embedding_f32(idx2, w, o)
    assert torch.equal(o[0], w[1]) and torch.equal(o[3], w[2])
    assert (o[1] == 0).all() and (o[2] == 0).all()
