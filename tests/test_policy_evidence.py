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


def test_tail_split_policy_against_the_committed_probe(built):
    """profiles/r04_hgemm_tail_probe_after.log (the policy's pick = the `plan` column, measured on the GPU box next to every tail candidate and
    the single-pass 256 x 256 kernel): at every size the tail plan takes, it is not slower than the single pass and within 3 % of the best
    candidate; and those sizes are exactly what cln_describe plans as a tail split among M = N = K = 256 ... 16384 step 256."""
    name = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    planned = {s for s in range(256, 16384 + 256, 256) if "tail split" in built.manifest.describe(name, (s, s, s), 2)}
    seen = set()
    for line in open(os.path.join(ROOT, "profiles", "r04_hgemm_tail_probe_after.log")):
        m = re.match(r"TAIL\s+(\d+)\^3 .*\| plan\s+([\d.]+) \(.*\) w4-256\s+([\d.]+) \| (.*)", line)
        size, plan, single = int(m.group(1)), float(m.group(2)), float(m.group(3))
        cands = [float(x) for x in re.findall(r"S=\d+\s+([\d.]+)", m.group(4))]
        seen.add(size)
        assert plan >= 0.995 * single, line
        assert plan >= 0.97 * max(cands), line
    assert seen == planned, (sorted(seen), sorted(planned))


def test_attention_small_grid_plan_against_the_committed_probe(built):
    """profiles/r04_fa_small_grid_probe.log: at all 120 shapes (D = 32 ... 256, N = 1024 / 2048 / 4096, 32 ... 256 blocks of 256 rows) the kernel
    fa2_plan picks was measured within 4 % of the best candidate, and cln_describe still names that kernel."""
    name = "flash_attn_mma_stages_split_q_shared_qkv"
    n = 0
    for line in open(os.path.join(ROOT, "profiles", "r04_fa_small_grid_probe.log")):
        m = re.match(r"SMALLGRID D=\s*(\d+) N=\s*(\d+) BH=\s*(\d+) wgs256=\s*\d+\s+v2x2\s+(\S+)\s+v2x4\s+(\S+)\s+v2x8\s+(\S+)\s+m16x\s+(\S+) \| plan\s+([\d.]+) \((\w+) (\d) waves\)", line)
        D, N, BH = int(m.group(1)), int(m.group(2)), int(m.group(3))
        cands = [float(x) for x in m.group(4, 5, 6, 7) if x != "nan"]
        assert float(m.group(8)) >= 0.96 * max(cands), line
        what = built.manifest.describe(name, (1, BH, N, D), 2)
        assert what.startswith(m.group(9) + "<") and ("%s waves" % m.group(10)) in what, (line, what)
        n += 1
    assert n == 120


def test_sgemm_tile_plan_is_near_the_measured_best_form(built):
    """csrc/sgemm.hip sgemm_plan (through cln_describe, host only) against the committed sweep of the three tile forms of the LDS-DMA f32-MFMA kernel
    (profiles/r06_sgemm_dma_sweep.log: the reference's sgemm sweep + ten off-sweep shapes, each form measured on one box): the planned form is within
    2.5 % of the best measured form at every shape, within 1.2 % on the reference's own sweep."""
    m = built.manifest
    log = os.path.join(ROOT, "profiles", "r06_sgemm_dma_sweep.log")
    rows = 0
    for ln in open(log):
        if not ln.startswith("SWEEP"):
            continue
        t = ln.split()
        M, N, K = int(t[1]), int(t[2]), int(t[3])
        forms = {t[i]: float(t[i + 1]) for i in range(9, len(t) - 3, 2) if t[i] != "|" and t[i + 1] != "n/a"}
        txt = m.describe("sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages", (M, N, K), 2)
        form = re.search(r"sgemm_dma<(\d+x\d+)x16", txt).group(1)
        assert form in forms, (form, forms)
        best = max(forms.values())
        in_sweep = M >= 4096 and N >= 4096 and K >= 2048
        assert forms[form] >= (0.988 if in_sweep else 0.975) * best, (M, N, K, form, forms)
        rows += 1
    assert rows >= 30
    assert not m.stages_honoured("sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages_dsmem", (4096, 4096, 4096), 3)
