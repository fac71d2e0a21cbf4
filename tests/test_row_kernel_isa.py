"""The softmax / layer-norm / rms-norm row kernels are VALU-co-limited at fp16 (8192^2 rates followed the VALU count per lane), so
the exact-fit instantiations -- a row that fills every pack of every lane, i.e. every power-of-two row length of the reference scripts
(kernels/softmax/softmax.py, layer-norm/layer_norm.py, rms-norm/rms_norm.py: S = 4096, K = 256 ... 8192) from 512 up -- must stay free of the per-pack
bounds test. Checked on the assembly hipcc emits for the product sources (no GPU needed)."""
import os
import re
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-learn-notes_amd", "tools"))


def _valu(src, want):
    import kernel_resources as kr
    kernels, s = kr.report(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", src))
    text = open(s).read()
    k = [k for k in kernels if want in k["demangled"].replace("_Float16", "half")]
    assert len(k) == 1, (want, [x["demangled"] for x in kernels][:4])
    m = re.search(r"^%s:" % re.escape(k[0]["name"]), text, re.M)
    body = text[m.end():text.index("s_endpgm", m.end())]
    ins = [ln.split(";")[0].strip().split()[0] for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((".", ";")) and not ln.strip().endswith(":")]
    c = Counter(ins)
    return sum(n for i, n in c.items() if i.startswith("v_")), c, k[0]


def test_exact_fit_row_kernels_carry_no_bounds_test():
    # 64 elements per lane (8 packs of 8 halves). Guarded forms measured 733 / 604 / 409 VALU instructions; the exact-fit forms 593 / 463 / 332.
    for src, full, guarded, budget in (("softmax.hip", "softmax_row_kernel<half, 8, 8, 1, true>", "softmax_row_kernel<half, 8, 8, 1, false>", 620),
                                       ("norm.hip", "layer_norm_kernel<half, 8, 8, true>", "layer_norm_kernel<half, 8, 8, false>", 490),
                                       ("norm.hip", "rms_norm_kernel<half, 8, 8, true>", "rms_norm_kernel<half, 8, 8, false>", 360)):
        n_full, c_full, k = _valu(src, full)
        n_guard, _, _ = _valu(src, guarded)
        assert n_full <= budget and n_full < 0.9 * n_guard, (full, n_full, n_guard)
        assert c_full["v_cndmask_b32_e32"] + c_full["v_cndmask_b32_e64"] <= 12, (full, c_full)  # the reductions' own selects only
        assert k["spill"] == 0 and k["scratch"] == 0


def test_fp16_softmax_forms_its_exponent_with_one_fma():
    _, c, _ = _valu("softmax.hip", "softmax_row_kernel<half, 8, 8, 1, true>")
    assert c["v_exp_f32_e32"] == 64 and c["v_sub_f32_e32"] <= 8, c  # (was: 64 v_sub + 64 v_mul in front of the 64 exponentials)
    # the fp32 rungs keep the reference's expf(x - max): subtract, then exponentiate
    _, c32, _ = _valu("softmax.hip", "softmax_row_kernel<float, 4, 8, 1, true>")
    assert c32["v_sub_f32_e32"] >= 32, c32
