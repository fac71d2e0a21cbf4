"""CPU: the oracle restatement agrees with the outputs of the REFERENCE's own Python functions
(fixtures written by tests/golden/make_golden.py, which executes the reference code)."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def seeded(seed, *shape, dtype=torch.float32):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(*shape, generator=g).to(dtype)


def test_attention_matches_reference_function(oracle):
    z = np.load(os.path.join(GOLD, "attn_b2h2n128d64.npz"))
    B, H, N, D = z["shape"]
    q, k, v = (seeded(s, B, H, N, D, dtype=torch.float16) for s in z["seeds"])
    ref = torch.from_numpy(z["out"])
    got = oracle.unfused_standard_attn(q.float(), k.float(), v.float())
    assert torch.allclose(got, ref, atol=1e-6, rtol=1e-6)
    # the fp64 oracle used by the GPU tests is the same function at higher precision
    assert (oracle.attention_fp64(q, k, v).float() - ref).abs().max() < 1e-5
    assert (oracle.sdpa(q.float(), k.float(), v.float()) - ref).abs().max() < 1e-5


def test_hgemm_matches_script_column(oracle):
    z = np.load(os.path.join(GOLD, "hgemm_128x192x256.npz"))
    M, N, K = z["shape"]
    a, b = seeded(z["seeds"][0], M, K, dtype=torch.float16), seeded(z["seeds"][1], K, N, dtype=torch.float16)
    ref = torch.from_numpy(z["out"])  # torch.matmul on fp16 (hgemm.py:420-421)
    # torch.matmul on fp16 CPU tensors accumulates in an order that depends on the host's ISA (AVX-512 / AMX / thread count): on the machine that
    # wrote the fixture the restatement is bit-equal; on another one a handful of results that sit on a rounding boundary land on the neighbouring
    # fp16 value (seen: 31 of 24 576 elements, 1 ulp). The bar that holds on every host: never more than one fp16 ulp (or the fp32 accumulation-order noise where the terms cancel), and < 0.5 % of the elements.
    got = oracle.hgemm_fp16_path(a, b)
    diff = (got.float() - ref.float()).abs()
    ulp = torch.maximum(ref.float().abs(), got.float().abs()).clamp_min(2.0 ** -14).log2().floor().exp2() * 2.0 ** -10
    order_noise = (a.float().abs() @ b.float().abs()) * 2.0 ** -20  # fp32 accumulation in another order: shows where the terms cancel (|C| ~ 1e-3)
    assert bool((diff <= torch.maximum(ulp, order_noise)).all()) and int((diff > 0).sum()) < 0.005 * diff.numel()
    # fp32-accumulate oracle differs from the fp16 script column by at most fp16 rounding of |C|~16
    assert (oracle.hgemm(a, b).float() - ref.float()).abs().max() <= 0.0625
    assert torch.equal(oracle.as_col_major(b), torch.from_numpy(z["b_col_major"]))
    # as_col_major keeps the [K,N] shape with [N,K] storage
    assert torch.equal(oracle.as_col_major(b).reshape(N, K).t().contiguous(), b)


def test_scalars(oracle):
    sc = json.load(open(os.path.join(GOLD, "scalars.json")))
    for key, val in sc["block_swizzle_stride"].items():
        n, k = map(int, key.split("_"))
        assert oracle.make_block_swizzle_stride(n, k) == val
    assert abs(oracle.get_mha_tflops(4, 8, 2048, 64, 1.0) - sc["mha_tflops_at_1s"]["C4"]) < 1e-12
    assert abs(oracle.get_mha_tflops(1, 32, 4096, 512, 1.0) - sc["mha_tflops_at_1s"]["C5"]) < 1e-12
    assert abs(oracle.get_mha_tflops(4, 8, 2048, 64, 1.0, True) - sc["mha_tflops_at_1s"]["C4_mm"]) < 1e-12


def test_row_ops_match_reference_functions(oracle):
    z = np.load(os.path.join(GOLD, "rows_64x512.npz"))
    x = seeded(z["seed"], 64, 512)
    t = lambda k: torch.from_numpy(z[k])
    assert torch.allclose(oracle.layer_norm_torch(x, 1.0, 0.0), t("layer_norm"), atol=2e-6)
    assert torch.allclose(oracle.layer_norm_torch(x, 1.5, -0.25), t("layer_norm_gb"), atol=2e-6)
    assert torch.allclose(oracle.rms_norm_torch(x, 1.0), t("rms_norm"), atol=2e-6)
    assert torch.allclose(oracle.rms_norm_torch(x, 0.5), t("rms_norm_g"), atol=2e-6)
    assert torch.allclose(oracle.rope_torch(x), t("rope"), atol=1e-6)
    assert torch.allclose(oracle.softmax_per_token(x), t("softmax"), atol=1e-7)
    assert torch.allclose(oracle.softmax_global(x).flatten(), t("softmax_global"), atol=1e-9)
    assert torch.equal(oracle.elementwise_add(x, seeded(301, 64, 512)), t("add"))
    assert abs(oracle.reduce_sum(x) - float(z["sum"])) < 1e-9


def test_kernel_semantics_vs_torch_oracle(oracle):
    """The documented quirks: kernel-semantics restatements stay within fp16 tolerance of the torch
    oracle for the norms, and rope's kernel quirk equals the torch oracle only at position 0."""
    x = seeded(7, 32, 1024)
    d = (oracle.layer_norm_kernel(x, 1.0, 0.0) - oracle.layer_norm_torch(x, 1.0, 0.0)).abs().max()
    assert d < 3e-3  # sqrt(1023/1024) factor on |y| <~ 4
    d = (oracle.rms_norm_kernel(x, 1.0) - oracle.rms_norm_torch(x, 1.0)).abs().max()
    assert d < 1e-4
    rk, rt = oracle.rope_kernel(x), oracle.rope_torch(x)
    assert torch.allclose(rk[0], rt[0], atol=1e-6)
    assert torch.allclose(rk[:, :2], rt[:, :2], atol=1e-4)  # pair 0 has frequency 1 in both
    assert (rk[5] - rt[5]).abs().max() > 0.1


def test_histogram_known_answer_from_reference_readme(oracle):
    """kernels/histogram/histogram.py:22 feeds list(range(10))*1000; the README transcript of the reference
    kernels (kernels/histogram/README.md:24-44) prints 1000 for each of the ten bins."""
    a = torch.tensor(list(range(10)) * 1000, dtype=torch.int32)
    h = oracle.histogram(a)
    assert h.dtype == torch.int32 and h.tolist() == [1000] * 10


def test_embedding_is_a_row_gather(oracle):
    g = torch.Generator().manual_seed(5)
    w = torch.randn(37, 24, generator=g)
    idx = torch.tensor([0, 36, 5, 5, 17], dtype=torch.int32)
    out = oracle.embedding(idx, w)
    for r, i in enumerate(idx.tolist()):
        assert torch.equal(out[r], w[i])


def test_kernel_restatement_of_layer_norm_is_tied_to_the_pinned_script_oracle(oracle):
    """`layer_norm_kernel` (the CUDA kernel's arithmetic, layer_norm.cu:53-72: population variance with eps added to
    K) cannot be run here, but it differs from the golden-pinned script function `layer_norm_torch` (unbiased std, no
    eps) by ONE analytic factor: y_kernel - b = (y_script - b) * sqrt((K + 1e-5) / (K - 1)). Checking that identity ties
    the restatement the GPU tests assert tightly against to the pinned oracle."""
    import math
    import torch
    g, b = 1.5, -0.25
    for K in (64, 1000, 4096):
        x = torch.randn(7, K, generator=torch.Generator().manual_seed(K)) * 2 + 0.5
        yk, yt = oracle.layer_norm_kernel(x, g, b), oracle.layer_norm_torch(x, g, b)
        derived = b + (yt - b) * math.sqrt((K + 1e-5) / (K - 1))
        assert torch.allclose(yk, derived, atol=2e-6, rtol=1e-6), K


def test_plain_attention_names_are_held_to_the_reference_plain_arithmetic(oracle):
    """Where the 6e-3 of tests/test_gpu_flash_attn.py (TOL_AMPLIFIED_KEYS, plain names at D <= 128) comes from. The reference's plain
    kernels accumulate Q K^T and P V in fp16 (mma.sync ... f16.f16.f16.f16, flash_attn_mma_share_qkv.cu:346, :555); emulated at their best
    (oracle.attention_reference_plain_arithmetic) they reach 4e-4 on N(0,1) inputs -- the README's "< 1e-3" -- and 3.5-4.0e-3 on the
    rescale-regime inputs of the GPU tests (keys amplified 3-5x, logits of +-40): ABOVE 3e-3, below 6e-3. So 3e-3 on those inputs is a bar
    the reference's own plain arithmetic does not meet; the plain names here (fp32 accumulation, Q * log2(e)/sqrt(d) rounded to fp16 once:
    2.7-4.3e-3 measured) are held to the bound that arithmetic class gives, the `*_acc_f32` names (the reference's precise rung) to 3e-3."""
    TOL, TOL_AMPLIFIED_KEYS = 3e-3, 6e-3
    B, H, N, D = 1, 3, 1024, 64
    g = lambda s: torch.randn(B, H, N, D, generator=torch.Generator().manual_seed(s)).half()  # noqa: E731 -- seeded() of the GPU test
    q, k, v = g(31), g(32), g(33)
    ramp = torch.linspace(0.2, 1.6, N).view(N, 1)
    k[0, 0] = (k[0, 0].float() * ramp).half()
    k[0, 0, 900] = q[0, 0, 5] * 3.0
    k[0, 1, 10] = q[0, 1, 300] * 5.0
    k[0, H - 1, 1000] = q[0, H - 1, 1023] * 4.0
    worst = 0.0
    for h in range(H):
        ref = oracle.attention_fp64(q[:, h:h + 1], k[:, h:h + 1], v[:, h:h + 1])[0, 0]
        o = oracle.attention_reference_plain_arithmetic(q[0, h], k[0, h], v[0, h])
        worst = max(worst, (o.double() - ref).abs().max().item())
    assert TOL < worst <= TOL_AMPLIFIED_KEYS, worst
    q, k, v = g(1), g(2), g(3)  # N(0,1): the plain arithmetic is well inside TOL (and the README's 1e-3)
    ref = oracle.attention_fp64(q[:, :1], k[:, :1], v[:, :1])[0, 0]
    err = (oracle.attention_reference_plain_arithmetic(q[0, 0], k[0, 0], v[0, 0]).double() - ref).abs().max().item()
    assert err < 1e-3, err
