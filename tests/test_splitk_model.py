"""CPU: the split-K time model compiled into csrc/hgemm.hip (splitk_plan) is the least-squares fit of the committed measurements, and what the
library plans (through cln_describe) is, where that candidate was measured, within 8 % of the measured best."""
import importlib.util
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "profiles", "r04_hgemm_splitk_probe.log")


def _tool():
    spec = importlib.util.spec_from_file_location("fit_splitk_model", os.path.join(ROOT, "cuda-learn-notes_amd", "tools", "fit_splitk_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_shipped_constants_are_the_fit_of_the_committed_sweep():
    t = _tool()
    rows = t.load(LOG)
    assert len(rows) >= 180 and len({r[:3] for r in rows}) == 28
    x = t.fit(rows)
    assert np.allclose(x, t.SHIPPED, rtol=0.01), (x, t.SHIPPED)
    res = np.array([np.log(t.model(t.SHIPPED, r) / r[6]) for r in rows])
    assert res.std() < 0.08
    assert max(loss for _, _, _, loss in t.picks(t.SHIPPED, rows)) < 0.08
    src = open(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", "hgemm.hip")).read()
    for c in t.SHIPPED:  # the same numbers, literally, in the planner
        assert ("%g" % c) in src, c


def test_library_plan_is_near_the_measured_best(built):
    t = _tool()
    by = {}
    for r in t.load(LOG):
        by.setdefault(r[:3], {})[(r[3], r[4], r[5])] = r[6]
    name = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    planned = 0
    for (M, N, K), cands in by.items():
        what = built.manifest.describe(name, (M, N, K), 2)
        m = re.match(r"hgemm_w4<(\d+)x(\d+)x64.* split-K x (\d+) ", what)
        inside = K >= 4096 and M * N <= 2048 * 2048 and (M * N <= 1536 * 1536 or K >= 5120)
        assert bool(m) == inside, (M, N, K, what)
        if m:
            key = tuple(int(g) for g in m.groups())
            if key in cands:  # (an unmeasured S, e.g. 7, is allowed: the sweep sampled S)
                planned += 1
                assert cands[key] <= 1.08 * min(cands.values()), ((M, N, K), key, cands[key], min(cands.values()))
    assert planned >= 15
