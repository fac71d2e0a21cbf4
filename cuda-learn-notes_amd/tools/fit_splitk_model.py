"""Fit of the split-K time model in csrc/hgemm.hip splitk_plan to the measured candidates (SPLITKALL lines of
profiles/r04_hgemm_splitk_probe.log, written by tools/hg_splitk_probe.py on the GPU box).  CPU only.

    t(tile, S) = rounds x (K / (64 S) x tau + phi x sqrt(BM BN / 256^2)) + rho0 + (4 S M N + 2 M N) / bw
    tau = 2 BM BN 64 / (5.86 TF x eff[tile]) x (1 - alpha (1 - fill)),  fill = min(tiles S, 256) / 256,  rounds = ceil(tiles S / 256)

python fit_splitk_model.py [log] -> the coefficients, the rms / max log-error, and per shape the model's pick against the measured best."""
import os
import re
import sys
from collections import defaultdict

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TILES = [(256, 256), (192, 256), (192, 192), (128, 256), (160, 160)]  # the shapes splitk_plan offers (256 x 128 was measured and dropped)
# the constants compiled into csrc/hgemm.hip: eff per tile (order of TILES), phi [us], rho0 [us], bw [TB/s], alpha
SHIPPED = [1.249, 1.188, 1.078, 0.974, 0.942, 3.88, 5.18, 3.615, 0.338]


def load(path):
    rows = []
    for line in open(path):
        if not line.startswith("SPLITKALL"):
            continue
        p = line.split()
        M, N, K = map(int, p[1:4])
        for kv in p[4:]:
            m = re.match(r"(\d+)x(\d+)S=(\d+)=([\d.]+)", kv)
            if m and (int(m[1]), int(m[2])) in TILES:
                bm, bn, S, tf = int(m[1]), int(m[2]), int(m[3]), float(m[4])
                rows.append((M, N, K, bm, bn, S, 2.0 * M * N * K / tf * 1e-6))  # measured time in us
    return rows


def model(x, r):
    M, N, K, bm, bn, S, _ = r
    eff, phi, rho0, bw, alpha = x[TILES.index((bm, bn))], x[5], x[6], x[7], x[8]
    n = (M // bm) * (N // bn) * S
    rounds = -(-n // 256)
    fill = min(n, 256) / 256.0
    tau = 2.0 * bm * bn * 64 / (5.86e6 * eff) * (1.0 - alpha * (1.0 - fill))
    return rounds * ((K // S // 64) * tau + phi * (bm * bn / 65536.0) ** 0.5) + rho0 + (4.0 * S * M * N + 2.0 * M * N) / (bw * 1e6)


def fit(rows):
    x0 = [1.0, 0.96, 0.93, 0.82, 0.88, 5.0, 4.0, 3.0, 0.1]
    lo, hi = [0.3] * 5 + [0, 0, 0.3, 0], [1.5] * 5 + [30, 30, 20, 0.9]
    return least_squares(lambda x: [np.log(model(x, r) / r[6]) for r in rows], x0, bounds=(lo, hi)).x


def picks(x, rows):
    by = defaultdict(list)
    for r in rows:
        by[r[:3]].append(r)
    out = []
    for shape, v in by.items():
        pred, best = min(v, key=lambda r: model(x, r)), min(v, key=lambda r: r[6])
        out.append((shape, pred[3:6], best[3:6], pred[6] / best[6] - 1.0))
    return out


if __name__ == "__main__":
    rows = load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_hgemm_splitk_probe.log"))
    x = fit(rows)
    res = np.array([np.log(model(x, r) / r[6]) for r in rows])
    print("candidates", len(rows), "fitted", np.round(x, 3).tolist(), "rms %.3f max %.3f" % (res.std(), abs(res).max()))
    res = np.array([np.log(model(SHIPPED, r) / r[6]) for r in rows])
    print("shipped  ", SHIPPED, "rms %.3f max %.3f" % (res.std(), abs(res).max()))
    for shape, pred, best, loss in picks(SHIPPED, rows):
        print("%-22s model picks %dx%d S=%-2d  measured best %dx%d S=%-2d  loss %.1f %%" % ((str(shape),) + tuple(pred) + tuple(best) + (loss * 100,)))
