"""GPU parity: the 42 activation entry points through the C-ABI vs the fp64 oracle.
Tolerance: fp32 rungs 2e-6 relative + 1e-6 absolute (fast __expf / tanhf), fp16 rungs one fp16 rounding of the
exact value (rel 1e-3); relu and hardshrink are selections of the input and must be BIT-EXACT."""
import pytest
import torch

pytestmark = pytest.mark.gpu
OPS = ("relu", "sigmoid", "gelu", "swish", "elu", "hardswish", "hardshrink")
RUNGS = ("f32", "f32x4", "f16", "f16x2", "f16x8", "f16x8_pack")


@pytest.fixture(scope="module")
def lib(built, dev):
    return built.load("activation")


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("shape", [(1024, 1024), (7, 24), (3, 5, 64)])
def test_activation_matches_oracle(lib, dev, oracle, op, shape):
    g = torch.Generator().manual_seed(hash(op) % 1000)
    x = torch.randn(*shape, generator=g) * 3.0
    x.view(-1)[:6] = torch.tensor([0.0, -0.0, 0.5, -0.5, 3.0, -3.0])  # thresholds of hardshrink / hardswish
    for rung in RUNGS:
        dt = torch.float32 if rung.startswith("f32") else torch.float16
        xd = x.to(dt)
        ref = oracle.activation(op, xd)
        y = torch.full(shape, 7.0, dtype=dt, device=dev)
        getattr(lib, "%s_%s" % (op, rung))(xd.to(dev), y)
        got = y.cpu().double()
        if op in ("relu", "hardshrink"):
            assert torch.equal(got, ref), (op, rung)
        else:
            tol = (2e-6, 1e-6) if dt == torch.float32 else (1e-3, 1e-4)
            assert torch.allclose(got, ref, rtol=tol[0], atol=tol[1]), (op, rung, (got - ref).abs().max().item())


def test_activation_extremes_are_finite(lib, dev):
    x = torch.tensor([-1e4, -100.0, -20.0, 20.0, 100.0, 1e4, 0.0, 1.0] * 8, device=dev)
    y = torch.zeros_like(x)
    for op in OPS:
        getattr(lib, op + "_f32x4")(x, y)
        assert torch.isfinite(y).all(), op
    xh, yh = x.clamp(-6e4, 6e4).half(), y.half()
    for op in OPS:
        getattr(lib, op + "_f16x8_pack")(xh, yh)
        assert torch.isfinite(yh).all(), op


def test_hardswish_edges_follow_the_reference_piecewise_form(lib, dev):
    """hardswish.cu:37-45: x >= 3 -> x EXACTLY (also for +inf and values above FLT_MAX / 6), x <= -3 -> 0 (also for -inf), NaN stays NaN
    (the branch-free x * med3(x + 3, 0, 6) / 6 gave NaN at -inf, +inf above 5.7e37 and 1 ulp off x for x >= 3)."""
    inf = float("inf")
    vals = [3.0, 3.0000002, 7.3, 1234.567, 1e38, 3.0e38, inf, -3.0, -3.0000002, -1e38, -inf, float("nan"), 0.0, -0.0, 1.0, -1.0]
    x = torch.tensor(vals * 4, device=dev)
    y = torch.full_like(x, 7.0)
    for rung in ("f32", "f32x4"):
        getattr(lib, "hardswish_" + rung)(x, y)
        got = y.cpu()[:len(vals)]
        for v, g in zip(vals, got.tolist()):
            if v != v:
                assert g != g
            elif v >= 3.0:
                assert g == torch.tensor(v, dtype=torch.float32).item(), (rung, v, g)
            elif v <= -3.0:
                assert g == 0.0, (rung, v, g)
        assert abs(got[14].item() - 4.0 / 6.0) < 1e-6 and abs(got[15].item() + 2.0 / 6.0) < 1e-6
    xh = torch.tensor([3.0, 100.0, 65504.0, inf, -3.0, -65504.0, -inf, 1.0] * 8, dtype=torch.half, device=dev)
    yh = torch.full_like(xh, 7.0)
    for rung in ("f16", "f16x2", "f16x8", "f16x8_pack"):
        getattr(lib, "hardswish_" + rung)(xh, yh)
        g = yh.cpu()[:8].tolist()
        assert g[:4] == [3.0, 100.0, 65504.0, inf] and g[4:7] == [0.0, 0.0, 0.0] and abs(g[7] - 4.0 / 6.0) < 1e-3, (rung, g)


def test_activation_dtype_error(lib, dev):
    x = torch.zeros(8, device=dev)
    with pytest.raises(RuntimeError, match="values must be torch::kHalf"):
        lib.gelu_f16x8_pack(x, x.clone())
