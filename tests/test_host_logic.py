"""CPU: host-side mirror behaviour -- argument checks, error texts, and loud failure without HIP."""
import os

import pytest
import torch


def test_cpu_tensors_are_rejected_loudly(built):
    ew = built.load("elementwise")
    a = torch.randn(4, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ew.elementwise_add_f32(a, a, a.clone())


def test_dtype_and_shape_errors_match_reference_texts(built):
    hg = built.hgemm_lib()
    a = torch.randn(128, 128)  # fp32 instead of half
    with pytest.raises(RuntimeError, match="values must be torch::kHalf"):
        hg.hgemm_naive_f16(a, a, a)
    fa = built.flash_attn_lib()
    q = torch.randn(1, 1, 128, 64)
    with pytest.raises(RuntimeError, match="values must be torch::kHalf"):
        fa.flash_attn_mma_stages_split_q_shared_qkv(q, q, q, q, 1)
    red = built.load("reduce")
    with pytest.raises(RuntimeError, match="values must be torch::kInt8"):
        red.block_all_reduce_sum_i8_i32(torch.randn(8))


def test_missing_library_raises_not_falls_back(built, monkeypatch, tmp_path):
    from cuda_learn_notes_amd import _loader
    monkeypatch.setattr(_loader, "LIBDIR", str(tmp_path))
    monkeypatch.setattr(_loader, "_cache", {})
    with pytest.raises(_loader.LibraryMissing, match="no CPU fallback"):
        _loader.load_so("libcln_amd.so")


def test_product_path_never_imports_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-learn-notes_amd")
    for dp, _, fns in os.walk(root):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cuh", ".h", ".inc")):
                src = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, fn
                assert "/root/reference" not in src, fn


def test_block_swizzle_policy_and_tflops_model(built):
    from cuda_learn_notes_amd import bench_utils as bu
    assert bu.make_block_swizzle_stride(4096, 4096) == 2048
    assert bu.make_block_swizzle_stride(256, 256) == 1
    assert abs(bu.hgemm_flops(4096, 4096, 4096) - 137438953472) < 1
    assert abs(bu.get_mha_tflops(4, 8, 2048, 64, 1.0) - 0.035026501632) < 1e-12


def test_bench_steady_state_check_stops_when_the_rate_settles(built, monkeypatch):
    """bench_utils.settle (the untimed check between the pre-warm and the timed region of bench.py): event-timed batches
    until the last three agree within tol and the last is within tol of the fastest; a slow plateau of two batches does
    not count as settled."""
    from cuda_learn_notes_amd import bench_utils as bu
    seq = iter([0.142, 0.139, 0.120, 0.096, 0.0945, 0.0950, 0.0947, 0.0946])  # a slow start that recovers (ms per launch)
    calls = []
    monkeypatch.setattr(bu, "time_region_events", lambda fn, n, stream=None: (calls.append(n), next(seq))[1])
    hist = bu.settle(lambda: None, 20, tol=0.03, max_seconds=5.0, min_seconds=0.0)
    assert hist == [0.142, 0.139, 0.120, 0.096, 0.0945, 0.0950]
    assert all(n == 20 for n in calls)


def test_bench_steady_state_check_is_bounded(built, monkeypatch):
    from cuda_learn_notes_amd import bench_utils as bu
    t = [0.0]

    def timed(fn, n, stream=None):
        t[0] += 1.0  # every batch "takes" a second and is 10 % slower than the one before: never settles
        return 0.1 * 1.1 ** t[0]

    monkeypatch.setattr(bu, "time_region_events", timed)
    monkeypatch.setattr(bu.time, "time", lambda: t[0])
    hist = bu.settle(lambda: None, 20, tol=0.001, max_seconds=4.0)
    assert len(hist) == 4
