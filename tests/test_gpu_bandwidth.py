"""GPU parity: elementwise / reduce / softmax / layer-norm / rms-norm / rope through the C-ABI vs the
CPU oracle on the same seeded inputs. Tolerances are stated per test."""
import os

import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def seeded(seed, *shape, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


@pytest.fixture(scope="module")
def lib(built, dev):
    return built.load("elementwise", "reduce", "softmax", "layer_norm", "rms_norm", "rope")


# ------------------------------------------------------------------ elementwise: bit-exact
@pytest.mark.parametrize("shape", [(2048, 2048), (1024, 1024), (7, 33), (1, 1), (3, 8), (4096, 1000)])
def test_elementwise_add_bit_exact(lib, dev, oracle, shape):
    a, b = seeded(1, *shape), seeded(2, *shape)
    ref32 = oracle.elementwise_add(a, b)
    for name in ("elementwise_add_f32", "elementwise_add_f32x4"):
        c = torch.zeros(shape, device=dev)
        getattr(lib, name)(a.to(dev), b.to(dev), c)
        assert torch.equal(c.cpu(), ref32), name
    ah, bh = a.half(), b.half()
    ref16 = oracle.elementwise_add(ah, bh)  # one rounding of the exact sum: identical on both sides
    for name in ("elementwise_add_f16", "elementwise_add_f16x2", "elementwise_add_f16x8",
                 "elementwise_add_f16x8_pack"):
        c = torch.zeros(shape, device=dev, dtype=torch.half)
        getattr(lib, name)(ah.to(dev), bh.to(dev), c)
        assert torch.equal(c.cpu(), ref16), name


def test_elementwise_linearity_full_size(lib, dev):
    """size-independent property at the C1 size and beyond: add(a,b) == add(b,a); add(a,0) == a."""
    a = torch.randn(4096, 4096, device=dev)
    b = torch.randn(4096, 4096, device=dev)
    c1, c2 = torch.empty_like(a), torch.empty_like(a)
    lib.elementwise_add_f32x4(a, b, c1)
    lib.elementwise_add_f32(b, a, c2)
    assert torch.equal(c1, c2)
    lib.elementwise_add_f32x4(a, torch.zeros_like(a), c1)
    assert torch.equal(c1, a)


# ------------------------------------------------------------------ reduce
REDUCE_F = ["f32_f32", "f32x4_f32", "f16_f16", "f16_f32", "f16x2_f16", "f16x2_f32", "f16x8_pack_f16",
            "f16x8_pack_f32", "bf16_bf16", "bf16_f32", "bf16x2_bf16", "bf16x2_f32", "bf16x8_pack_bf16",
            "bf16x8_pack_f32", "fp8_e4m3_f16", "fp8_e4m3x16_pack_f16", "fp8_e5m2_f16", "fp8_e5m2x16_pack_f16"]


@pytest.mark.parametrize("shape", [(1024, 1024), (257, 33), (5,), (4096, 2048)])
@pytest.mark.parametrize("variant", REDUCE_F)
def test_reduce_float(lib, dev, oracle, built, variant, shape):
    name = "block_all_reduce_sum_" + variant
    in_dt = getattr(torch, built.manifest.REDUCE_DTYPES[name][0])
    x = seeded(11, *shape).to(in_dt)
    exact = oracle.reduce_sum(x)  # fp64 sum of the stored values
    y = getattr(lib, name)(x.to(dev))
    assert y.dtype == torch.float32 and y.numel() == 1
    n = x.numel()
    # error model: fp32 accumulation of n values with |x|~1: random-walk rounding ~ sqrt(n)*eps32*|partial|
    # plus, for *_f16/_bf16 in-pack accumulation, one half-precision rounding per pack (<= 2^-11 / 2^-8 relative
    # of the pack sum). Atomic order is non-deterministic, hence a tolerance even for f32.
    sum_abs = float(x.to(torch.float64).abs().sum())
    tol = 1e-6 * sum_abs + 1e-3
    if variant.endswith(("_f16",)) and ("x" in variant.split("_")[0] or "pack" in variant):
        tol += 2 ** -10 * sum_abs
    if variant.endswith("_bf16") and ("x" in variant):
        tol += 2 ** -7 * sum_abs
    assert abs(y.item() - exact) <= tol, (y.item(), exact, tol)


@pytest.mark.parametrize("shape", [(1024, 1024), (4096, 4096), (3, 5, 7), (1,)])
def test_reduce_i8_bit_exact(lib, dev, oracle, shape):
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-128, 128, shape, generator=g, dtype=torch.int8)
    exact = oracle.reduce_sum(x)
    for name in ("block_all_reduce_sum_i8_i32", "block_all_reduce_sum_i8x16_pack_i32"):
        y = getattr(lib, name)(x.to(dev))
        assert y.dtype == torch.int32
        assert y.item() == exact, name


# ------------------------------------------------------------------ softmax
@pytest.mark.parametrize("H", [64, 256, 1024, 4096, 8192, 1000])
def test_softmax_per_token(lib, dev, oracle, H):
    S = 128
    x = seeded(21, S, H) * 3.0
    ref = oracle.softmax_per_token(x)  # fp64 math rounded to fp32
    for name in ("softmax_f32_per_token", "softmax_f32x4_per_token", "safe_softmax_f32_per_token",
                 "safe_softmax_f32x4_per_token", "online_safe_softmax_f32_per_token",
                 "online_safe_softmax_f32x4_pack_per_token"):
        y = torch.zeros(S, H, device=dev)
        getattr(lib, name)(x.to(dev), y)
        # fast-math exp (reference builds with --use_fast_math): 2e-6 abs on probabilities <= 1
        assert torch.allclose(y.cpu(), ref, atol=2e-6, rtol=2e-5), name
        assert torch.allclose(y.sum(dim=1).cpu(), torch.ones(S), atol=1e-5)
    if H % 8 == 0:
        xh = x.half()
        refh = oracle.softmax_per_token(xh)
        for name in ("safe_softmax_f16_f32_per_token", "safe_softmax_f16x2_f32_per_token",
                     "safe_softmax_f16x8_pack_f32_per_token"):
            y = torch.zeros(S, H, device=dev, dtype=torch.half)
            getattr(lib, name)(xh.to(dev), y)
            assert torch.allclose(y.cpu().float(), refh.float(), atol=1e-6, rtol=2e-3), name  # 1 fp16 ulp


def test_softmax_safe_handles_large_logits(lib, dev, oracle):
    x = seeded(22, 16, 512) * 50.0 + 200.0  # exp overflows fp32 without max subtraction
    ref = oracle.softmax_per_token(x)
    for name in ("safe_softmax_f32_per_token", "online_safe_softmax_f32x4_pack_per_token"):
        y = torch.zeros(16, 512, device=dev)
        getattr(lib, name)(x.to(dev), y)
        assert torch.isfinite(y).all()
        assert torch.allclose(y.cpu(), ref, atol=1e-5, rtol=1e-4), name


@pytest.mark.parametrize("n", [16384, 4096 * 256, 1000])
def test_softmax_global(lib, dev, oracle, n):
    x = seeded(23, n)
    ref = oracle.softmax_global(x)
    for name in ("softmax_f32", "softmax_f32x4"):
        if name.endswith("x4") and n % 4:
            continue
        y = torch.zeros(n, device=dev)
        getattr(lib, name)(x.to(dev), y)
        assert torch.allclose(y.cpu(), ref, rtol=2e-5, atol=1e-12), name
        assert abs(y.sum().item() - 1.0) < 1e-4


# ------------------------------------------------------------------ norms
LN_F32 = ["layer_norm_f32", "layer_norm_f32x4"]
LN_F16 = ["layer_norm_f16_f16", "layer_norm_f16_f32", "layer_norm_f16x2_f16", "layer_norm_f16x8_f16",
          "layer_norm_f16x8_pack_f16", "layer_norm_f16x8_pack_f32"]
RMS_F32 = ["rms_norm_f32", "rms_norm_f32x4"]
RMS_F16 = ["rms_norm_f16_f16", "rms_norm_f16x2_f16", "rms_norm_f16x8_f16", "rms_norm_f16x8_pack_f16",
           "rms_norm_f16x8_f32", "rms_norm_f16x8_pack_f32", "rms_norm_f16_f32"]


@pytest.mark.parametrize("K", [64, 512, 1024, 4096, 8192, 1000])
def test_layer_norm(lib, dev, oracle, K):
    N, g, b = 96, 1.5, -0.25
    x = seeded(31, N, K) * 2.0 + 0.5
    ref_k = oracle.layer_norm_kernel(x, g, b)  # the reference KERNEL's arithmetic (population var, eps on K)
    ref_t = oracle.layer_norm_torch(x, g, b)   # the script's torch column (unbiased std)
    for name in LN_F32:
        y = torch.zeros(N, K, device=dev)
        getattr(lib, name)(x.to(dev), y, g, b)
        assert torch.allclose(y.cpu(), ref_k, atol=2e-5, rtol=1e-5), name
        assert torch.allclose(y.cpu(), ref_t, atol=6.0 / K + 1e-4), name  # sqrt((K-1)/K) factor on |y| <~ 6
        # ... and TIGHT against the pinned script oracle once the analytic difference between the two definitions is
        # applied: the kernel divides sum((x-mean)^2) by (K + 1e-5), the script by (K - 1), nothing else differs
        derived = b + (ref_t - b) * math.sqrt((K + 1e-5) / (K - 1))
        assert torch.allclose(y.cpu(), derived, atol=3e-5, rtol=1e-5), name
    if K % 8 == 0:
        xh = x.half()
        ref_kh = oracle.layer_norm_kernel(xh, g, b)
        for name in LN_F16:
            y = torch.zeros(N, K, device=dev, dtype=torch.half)
            getattr(lib, name)(xh.to(dev), y, g, b)
            assert torch.allclose(y.cpu().float(), ref_kh.float(), atol=1e-3, rtol=2e-3), name  # 1 fp16 ulp


@pytest.mark.parametrize("K", [64, 512, 1024, 4096, 8192, 1000])
def test_rms_norm(lib, dev, oracle, K):
    N, g = 96, 0.75
    x = seeded(32, N, K) * 2.0
    ref_k = oracle.rms_norm_kernel(x, g)
    ref_t = oracle.rms_norm_torch(x, g)
    for name in RMS_F32:
        y = torch.zeros(N, K, device=dev)
        getattr(lib, name)(x.to(dev), y, g)
        assert torch.allclose(y.cpu(), ref_k, atol=2e-5, rtol=1e-5), name
        assert torch.allclose(y.cpu(), ref_t, atol=1e-4), name  # eps 1e-5 on mean(x^2) ~ 4
    if K % 8 == 0:
        xh = x.half()
        ref_kh = oracle.rms_norm_kernel(xh, g)
        for name in RMS_F16:
            y = torch.zeros(N, K, device=dev, dtype=torch.half)
            getattr(lib, name)(xh.to(dev), y, g)
            assert torch.allclose(y.cpu().float(), ref_kh.float(), atol=1e-3, rtol=2e-3), name


def test_norm_full_size_properties(lib, dev):
    """[4096,4096] / [8192,8192] fp16: rows of layer-norm have mean ~0 / var ~1, rms-norm rows have unit rms."""
    x = torch.randn(4096, 4096, device=dev) * 3 + 1
    y = torch.empty_like(x)
    lib.layer_norm_f32x4(x, y, 1.0, 0.0)
    assert y.mean(dim=1).abs().max() < 1e-4
    assert (y.var(dim=1, unbiased=False) - 1).abs().max() < 1e-3
    xh = torch.randn(8192, 8192, device=dev, dtype=torch.half)
    yh = torch.empty_like(xh)
    lib.rms_norm_f16x8_pack_f32(xh, yh, 1.0)
    rms = yh.float().pow(2).mean(dim=1).sqrt()
    assert (rms - 1).abs().max() < 2e-3


# ------------------------------------------------------------------ rope
@pytest.mark.parametrize("shape", [(64, 128), (4096, 512), (8192, 1024), (17, 64)])
def test_rope_matches_torch_oracle(lib, dev, oracle, shape):
    """Default mode = the script's torch semantics (rope.py:68-88), PINNED by tests/golden. freq is formed as the
    script forms it (1 / theta ** (2i/dim), fp32 pow): wherever the device pow and torch's CPU pow agree to the bit
    the outputs agree to 2e-4 at EVERY position; where they differ by an ulp (torch's vectorised CPU pow is itself
    off the correctly rounded value in ~1 % of the columns) the angle t * freq differs by t * ulp(freq), which is the
    only slack the bound below allows."""
    x = seeded(41, *shape)
    ref = oracle.rope_torch(x)
    S, Hd = shape
    t = torch.arange(S, dtype=torch.float64).view(S, 1)
    freq = (1.0 / (10000.0 ** (torch.arange(0, Hd, 2).float() / Hd))).double().view(1, Hd // 2)
    pair_norm = x.double().view(S, -1, 2).norm(dim=-1)
    # 2e-4 + |pair| * t * (4 ulp of freq: device pow <= 2 ulp, torch CPU pow <= 1 ulp, the division 1 more): 2e-4 alone
    # for the first positions, up to 6e-3 at t = 8192 in the first columns
    bound = (2e-4 + pair_norm * t * freq * 4.8e-7).repeat_interleave(2, dim=1)
    for name in ("rope_f32", "rope_f32_v2", "rope_f32x4_pack"):
        out = torch.zeros(shape, device=dev)
        getattr(lib, name)(x.to(dev), out)
        d = (out.cpu().double() - ref.double()).abs()
        assert bool((d <= bound).all()), (name, float((d - bound).max()))
        # most columns carry a bit-identical freq: there the error is 2e-4 at all positions
        colmax = d.max(dim=0).values
        assert float((colmax <= 2e-4).double().mean()) >= 0.75, (name, float(colmax.max()))
        # rotation preserves each pair's norm
        n_in = x.view(shape[0], -1, 2).norm(dim=-1)
        n_out = out.cpu().view(shape[0], -1, 2).norm(dim=-1)
        assert torch.allclose(n_in, n_out, atol=1e-4, rtol=1e-4)


def test_rope_reference_kernel_quirk_mode(lib, dev, oracle):
    """ref_quirk=True: the reference CUDA kernels' integer-division behaviour (every pair rotated by t radians,
    rope.cu:26); both modes are reachable per call, no process-wide switch needed."""
    x = seeded(42, 512, 256)
    ref = oracle.rope_kernel(x)
    for name in ("rope_f32", "rope_f32x4_pack"):
        out = torch.zeros(512, 256, device=dev)
        getattr(lib, name)(x.to(dev), out, ref_quirk=True)
        assert (out.cpu() - ref).abs().max() < 5e-4, name
        out2 = torch.zeros(512, 256, device=dev)
        getattr(lib, name)(x.to(dev), out2, ref_quirk=False)
        assert (out2.cpu() - oracle.rope_torch(x)).abs().max() < 2e-3, name


# ------------------------------------------------------------------ streaming (non-temporal) store path
def test_launches_past_the_mall_use_streaming_stores_and_give_the_same_bits(built, lib, dev):
    """A launch whose tensors exceed the 256 MB MALL together writes its output with non-temporal stores
    (csrc/common.h cln_stream_nt / cln_store_stream). Same arithmetic, so an [8192, 8192] fp32 launch (512 MB: the
    streaming path) must equal, bit for bit, the same rows computed as two [4096, 8192] launches (256 MB each: the plain
    path, which the tests above hold to the oracle). Every kernel family that has the path, every access width."""
    act = built.load("activation")
    S, H = 8192, 8192
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(S, H, device=dev, generator=g) * 2.0
    halves = ((0, S // 2), (S // 2, S))

    def check(call, dtype, names):
        xin = x.to(dtype)
        for name in names:
            big = torch.full((S, H), 7.0, device=dev, dtype=dtype)
            call(name, xin, big)
            for (a, b) in halves:
                part = torch.full((b - a, H), 7.0, device=dev, dtype=dtype)
                call(name, xin[a:b], part)
                assert torch.equal(big[a:b], part), name
            del big

    check(lambda n, i, o: getattr(lib, n)(i, o), torch.float32, ("safe_softmax_f32x4_per_token", "softmax_f32_per_token"))
    check(lambda n, i, o: getattr(lib, n)(i, o, 1.5, -0.25), torch.float32, ("layer_norm_f32x4", "layer_norm_f32"))
    check(lambda n, i, o: getattr(lib, n)(i, o, 1.5), torch.float32, ("rms_norm_f32x4", "rms_norm_f32"))
    check(lambda n, i, o: getattr(act, n)(i, o), torch.float32, ("gelu_f32x4", "relu_f32"))
    # fp16 tensors of this shape are 128 MB each: 256 MB per two-tensor launch = the threshold, streaming path as well
    check(lambda n, i, o: getattr(lib, n)(i, o), torch.float16, ("safe_softmax_f16x8_pack_f32_per_token", "safe_softmax_f16x2_f32_per_token"))
    check(lambda n, i, o: getattr(lib, n)(i, o, 1.5, -0.25), torch.float16, ("layer_norm_f16x8_pack_f32", "layer_norm_f16_f32"))
    check(lambda n, i, o: getattr(lib, n)(i, o, 1.5), torch.float16, ("rms_norm_f16x8_pack_f32", "rms_norm_f16x8_f32"))
    check(lambda n, i, o: getattr(act, n)(i, o), torch.float16, ("relu_f16x8_pack", "swish_f16x2", "elu_f16"))
    xh = x.half()
    yh = torch.randn(S, H, device=dev, generator=g).half()
    for name in ("elementwise_add_f16x8_pack", "elementwise_add_f16x8", "elementwise_add_f16x2", "elementwise_add_f16"):
        big = torch.zeros(S, H, device=dev, dtype=torch.half)
        getattr(lib, name)(xh, yh, big)
        assert torch.equal(big, xh + yh), name  # one rounding of the exact sum on both sides
    big = torch.zeros(S, H, device=dev)
    y32 = yh.float()
    lib.elementwise_add_f32x4(x, y32, big)
    assert torch.equal(big, x + y32)
    # rope: position-dependent, so the reference is the same kernel on the plain path in four row blocks ... of the SAME
    # positions, which the API cannot express; use the property instead: every pair keeps its norm, and row 0 is unchanged
    out = torch.zeros(S, H, device=dev)
    lib.rope_f32x4_pack(x, out)
    assert torch.equal(out[0], x[0])
    assert torch.allclose(out.view(S, -1, 2).norm(dim=-1), x.view(S, -1, 2).norm(dim=-1), atol=1e-4, rtol=1e-4)
    small = torch.zeros(2048, H, device=dev)
    lib.rope_f32x4_pack(x[:2048].contiguous(), small)  # 128 MB: plain path, same positions 0..2047
    assert torch.equal(out[:2048], small)


# ------------------------------------------------------------------ scalar results without a zeroed output (round 5)
def test_reduce_and_dot_overwrite_their_result(lib, built, dev, oracle):
    """csrc/stream_scratch.h: the launch OVERWRITES y (the last block moves the total out of a self-resetting per-stream scratch word), so the
    result tensor no longer has to be zeroed -- through the raw C-ABI with y pre-filled with garbage, repeatedly, on several streams at once,
    and through the Python wrappers (which now hand out torch.empty results)."""
    from cuda_learn_notes_amd import _loader, host
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1 << 20, generator=g)
    xd = x.to(dev)
    exact = oracle.reduce_sum(x)
    raw = _loader.symbol("block_all_reduce_sum_f32x4_f32")
    y = torch.full((1,), float("nan"), device=dev)
    for _ in range(5):
        y.fill_(float("nan"))
        assert raw(xd.data_ptr(), y.data_ptr(), xd.numel(), host._stream()) == 0
        assert abs(y.item() - exact) <= 1e-3 * max(1.0, abs(exact)) + 1e-2
    xi = torch.randint(-128, 128, (1 << 20,), dtype=torch.int8, generator=g)
    yi = torch.full((1,), 12345, dtype=torch.int32, device=dev)
    assert _loader.symbol("block_all_reduce_sum_i8x16_pack_i32")(xi.to(dev).data_ptr(), yi.data_ptr(), xi.numel(), host._stream()) == 0
    assert yi.item() == int(xi.to(torch.int64).sum().item())
    # several streams at once: one scratch slot per stream
    streams = [torch.cuda.Stream() for _ in range(4)]
    parts = [torch.randn(1 << 18, generator=g) for _ in range(4)]
    res = []
    for rep in range(6):
        for st, p in zip(streams, parts):
            with torch.cuda.stream(st):
                res.append((lib.block_all_reduce_sum_f32x4_f32(p.to(dev, non_blocking=False)), p))
    torch.cuda.synchronize()
    for r, p in res:
        e = oracle.reduce_sum(p)
        assert abs(r.item() - e) <= 1e-3 * max(1.0, abs(e)) + 1e-2
    # n == 0 still yields a zero
    z = torch.full((1,), 3.0, device=dev)
    assert raw(xd.data_ptr(), z.data_ptr(), 0, host._stream()) == 0
    torch.cuda.synchronize()
    assert z.item() == 0.0
    # dot product: same scheme
    d = built.load("dot_product")
    a, b = torch.randn(1 << 18, generator=g), torch.randn(1 << 18, generator=g)
    want = (a.double() * b.double()).sum().item()
    for _ in range(3):
        got = d.dot_prod_f32x4_f32(a.to(dev), b.to(dev))
        assert abs(got.item() - want) <= 1e-3 * max(1.0, abs(want)) + 1e-2


def test_captured_reduce_carries_no_library_state_replay_overlaps_eager_launches(lib, dev, oracle):
    """a stream that ALREADY owns a scratch slot is captured. The graph must not bake that slot in -- torch replays on the
    current stream, so a replay can overlap eager launches of the capture stream (or another replay), and a lost or doubled ticket in a shared slot
    never recovers. Round 6: a captured launch always takes the memset + atomicAdd form. Warm up on s, capture on s, then replay on a SECOND stream
    while eager launches run on s: every result right, and the eager path of s still right afterwards."""
    g = torch.Generator().manual_seed(16)
    xs = [torch.randn(1 << 20, generator=g) for _ in range(2)]
    xd = [x.to(dev) for x in xs]
    exact = [oracle.reduce_sum(x) for x in xs]
    tol = [1e-3 * max(1.0, abs(e)) + 1e-2 for e in exact]
    s, other = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            assert abs(lib.block_all_reduce_sum_f32x4_f32(xd[0]).item() - exact[0]) <= tol[0]  # s owns a slot now
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            y = lib.block_all_reduce_sum_f32x4_f32(xd[1])
    torch.cuda.synchronize()
    eager = []
    for rnd in range(50):
        with torch.cuda.stream(other):
            graph.replay()
            got = y.clone()
        with torch.cuda.stream(s):
            eager.append(lib.block_all_reduce_sum_f32x4_f32(xd[0]))
        other.synchronize()
        assert abs(got.item() - exact[1]) <= tol[1], rnd
    torch.cuda.synchronize()
    assert all(abs(e.item() - exact[0]) <= tol[0] for e in eager)
    with torch.cuda.stream(s):
        assert abs(lib.block_all_reduce_sum_f32x4_f32(xd[0]).item() - exact[0]) <= tol[0]


def test_reduce_under_stream_capture_takes_the_memset_path(lib, dev, oracle):
    """No scratch slot may be allocated while a stream is being captured: a stream without one zeroes y with a memset node and adds into it
    directly; replays give the same sum."""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1 << 18, generator=g)
    xd = x.to(dev)
    exact = oracle.reduce_sum(x)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            y = lib.block_all_reduce_sum_f32x4_f32(xd)
        for _ in range(3):
            graph.replay()
            st.synchronize()
            assert abs(y.item() - exact) <= 1e-3 * max(1.0, abs(exact)) + 1e-2
