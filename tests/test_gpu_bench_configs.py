"""GPU: the side rows of bench.py (bench_configs.py at the repository root) run and are well-formed -- every BASELINE config and the
bandwidth kernels reach the driver's JSON line through this code, so a regression here would silently drop them to `error` entries."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bc(built):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench_configs
    return bench_configs


def test_config_c1_rows(bc, dev, oracle):
    r = bc.config_c1(dev, oracle)
    assert r["shape"] == [2048, 2048] and r["algorithmic_bytes"] == 3 * 4 * 2048 * 2048
    assert r["bit_exact_vs_torch_cpu"] is True
    for k in ("elementwise_add_f32", "elementwise_add_f32x4"):
        assert 500.0 < r[k]["gbps"] < 8000.0 and r[k]["rotating_sets"] >= 3, r[k]  # a rate, below the HBM spec peak
        assert r[k]["gbps_same_buffers"] > 0
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "GB/s" and cb["value"] > 0 and cb["cores"] >= 1


def test_bandwidth_rows_small(bc, dev, oracle, monkeypatch):
    monkeypatch.setattr(bc, "ROTATE_FOOTPRINT", 16 << 20)
    rows = bc.bandwidth_rows(dev, oracle, shapes=((1024, 1024),), cpu_shape=(1024, 1024))
    assert len(rows) == 11 and not [r for r in rows if "error" in r], rows
    for r in rows:
        bpe = {"elementwise_add_f32x4": 12, "elementwise_add_f16x8_pack": 6, "block_all_reduce_sum_f32x4_f32": 4,
               "block_all_reduce_sum_f16x8_pack_f32": 2}.get(r["kernel"], 8 if r["dtype"] == "f32" else 4)
        assert r["algorithmic_bytes"] == bpe * 1024 * 1024, r  # SURVEY 8(d) bytes per element
        assert r["rotating_sets"] >= 3 and r["launches"] >= 3 * r["rotating_sets"] and r["gbps"] > 0
        assert abs(r["frac_of_8TBs"] - r["gbps"] / 8000.0) < 1e-3
        assert r["cpu_baseline"]["value"] > 0 and r["cpu_baseline"]["kind"] == "port"


def test_hgemm_rows_small(bc, built, dev, oracle):
    r = bc.hgemm_config_rows(built, dev, oracle, sizes=(2048,), stage_list=(2, 3))
    h = r["hgemm_2048"]
    assert h["flops"] == 2.0 * 2048 ** 3
    for k in ("nn_stages2", "nn_stages3", "tn_stages2", "tn_stages3"):
        assert 50.0 < h[k]["tflops"] < 2500.0 and h[k]["kernel"], (k, h[k])
    assert "rocblas_nn_tflops" in h and "pct_of_rocblas_nn" in h["nn_stages2"]
    c2 = r["hgemm_c2_1024"]
    assert c2["naive"]["tflops"] > 1 and c2["mma2x4_warp4x4"]["tflops"] > c2["naive"]["tflops"]
    assert c2["cpu_baseline"]["kind"] == "port" and c2["max_abs_err_vs_cpu_fp16_matmul"] <= 0.25


def test_hgemm_policy_rows(bc, built, dev):
    r = bc.hgemm_policy_rows(built, dev, shapes=((512, 512, 8192), (4352, 4352, 4352)))
    sk, tail = r["512x512x8192"], r["4352x4352x4352"]
    assert "split-K x " in sk["plan"] and "tail split" in tail["plan"], (sk["plan"], tail["plan"])
    for row in (sk, tail):
        assert row["sampled_rows_within_one_fp16_ulp"] is True and 20.0 < row["nn_tflops"] < 2500.0, row
        assert "rocblas_nn_tflops" in row and "pct_of_rocblas_nn" in row, row


def test_fa_stage_rows(bc, built, dev):
    r = bc.fa_stage_rows(built, dev)
    for key, one_over_two in (("fa2_c4_d64", 0.8), ("fa2_d128", 0.8), ("fa2_c5_d512", 0.6)):
        row = r[key]
        assert row["stage1_bit_identical_to_stage2"] is True, key
        assert row["stage1_over_stage2"] >= one_over_two, (key, row["stage1_over_stage2"])  # C4 stages = 1 >= 0.8x stages = 2
        assert "single stage" in row["stages1"]["kernel"] and "single stage" not in row["stages2"]["kernel"]
