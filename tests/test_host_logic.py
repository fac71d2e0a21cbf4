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
                assert "load_oracle" not in src, fn  # (the bench's cpu_baseline legs live at the repository root: bench.py, bench_configs.py)
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


def test_interleaved_tile_walk_is_a_bijection_with_compact_xcd_chunks():
    """Python model of csrc/hgemm_mfma.cuh tile_coords_interleaved (the walk of the one-wave-per-SIMD HGEMM when A + B exceed twice the
    Infinity Cache): every workgroup index maps to a distinct tile for ragged and exact grids and any band; in every full round the 32 tiles
    of an XCD are 32 CONSECUTIVE tiles of the band walk (a 4 x 8 sub-block when the band is 8 tile columns wide), and the 256 tiles of a
    round are consecutive too (32 rows x 8 columns: 8 B panels shared by all XCDs)."""
    def coords(bid, nblk, tiles_m, tiles_n, band):
        full = nblk & ~255
        wg = bid
        if bid < full:
            xcd, local = bid & 7, bid >> 3
            wg = ((local >> 5) << 8) + (xcd << 5) + (local & 31)
        if band <= 0 or band > tiles_n:
            band = tiles_n
        per_band = tiles_m * band
        b = wg // per_band
        rem = wg - b * per_band
        width = min(band, tiles_n - b * band)
        tm = rem // width
        return wg, tm, b * band + (rem - tm * width)

    for tiles_m, tiles_n, band in ((64, 64, 8), (49, 49, 12), (60, 60, 7), (33, 32, 8), (40, 40, 10), (17, 70, 8), (64, 64, 0), (5, 7, 3)):
        nblk = tiles_m * tiles_n
        seen = {}
        for bid in range(nblk):
            wg, tm, tn = coords(bid, nblk, tiles_m, tiles_n, band)
            assert 0 <= tm < tiles_m and 0 <= tn < tiles_n, (tiles_m, tiles_n, band, bid)
            assert (tm, tn) not in seen, (tiles_m, tiles_n, band, bid, seen[(tm, tn)])
            seen[(tm, tn)] = bid
        assert len(seen) == nblk
        for r in range(nblk // 256):  # a full round: bids 256 r ... 256 r + 255 (the hardware places bid on XCD bid % 8)
            wgs = sorted(coords(b, nblk, tiles_m, tiles_n, band)[0] for b in range(256 * r, 256 * r + 256))
            assert wgs == list(range(256 * r, 256 * r + 256))
            for x in range(8):
                mine = sorted(coords(b, nblk, tiles_m, tiles_n, band)[0] for b in range(256 * r, 256 * r + 256) if b % 8 == x)
                assert mine == list(range(256 * r + 32 * x, 256 * r + 32 * x + 32)), (r, x)
    # band of 8 columns on an exact grid: an XCD's chunk is 4 rows x 8 columns
    tiles = [coords(b, 4096, 64, 64, 8)[1:] for b in range(256) if b % 8 == 3]
    assert len({t[0] for t in tiles}) == 4 and len({t[1] for t in tiles}) == 8


def test_split_k_workspace_layout_and_reduce_mapping_cover_the_tile_exactly_once():
    """Python model of csrc/hgemm_w4.cuh (EPI 5 partial store) + csrc/hgemm_splitk.cuh (hgemm_splitk_reduce_kernel): lane l of fragment (i, j) of
    wave w holds row 16 i + (l & 15), columns 16 j + 4 (l >> 4) ... + 3 of the wave tile (the layout store_wide_tile_via_lds reads); the GEMM writes
    those four floats at ((w FM FN + i FN + j) 64 + l) 4 of the tile's workspace block; reduce workgroup (tile, w, i) reads them back with the same
    index, puts them into its 16 x WTN strip and writes 16-byte row segments. For every tile shape split-K uses: each workspace float lands on
    exactly one element of C, every element of the tile is written exactly once, by a 16-byte-aligned 8-half store."""
    import numpy as np
    for BM, BN in ((256, 256), (192, 256), (192, 192), (128, 256), (160, 160)):
        FM, FN, WTM, WTN = BM // 32, BN // 32, BM // 2, BN // 2
        # the value each workspace slot would hold if the accumulators held their own (row, col) index
        ws = np.full(BM * BN, -1, dtype=np.int64)
        for w in range(4):
            wm, wn = w >> 1, w & 1
            for i in range(FM):
                for j in range(FN):
                    for l in range(64):
                        for e in range(4):
                            row, col = wm * WTM + i * 16 + (l & 15), wn * WTN + j * 16 + 4 * (l >> 4) + e
                            slot = w * (WTM * WTN) + ((i * FN + j) * 64 + l) * 4 + e
                            assert ws[slot] == -1
                            ws[slot] = row * BN + col
        assert (ws >= 0).all() and len(set(ws.tolist())) == BM * BN  # a bijection tile <-> workspace block
        out = np.full((BM, BN), -1, dtype=np.int64)
        RS, LPR = FN * 32 + 16, FN * 2
        for w in range(4):
            for i in range(FM):  # one reduce workgroup
                strip = {}
                for t in range(256):
                    wv, lane = t >> 6, t & 63
                    for j in range(wv, FN, 4):
                        base = w * (WTM * WTN) + ((i * FN + j) * 64 + lane) * 4
                        for e in range(4):
                            strip[(lane & 15) * RS + (j * 16 + 4 * (lane >> 4) + e) * 2] = ws[base + e]
                for t in range(16 * LPR):
                    r, c = t // LPR, t % LPR
                    row, col0 = (w >> 1) * WTM + i * 16 + r, (w & 1) * WTN + c * 8
                    assert col0 % 8 == 0
                    for e in range(8):
                        assert out[row, col0 + e] == -1
                        out[row, col0 + e] = strip[r * RS + c * 16 + e * 2]
        assert (out == np.arange(BM * BN).reshape(BM, BN)).all(), (BM, BN)


def test_global_softmax_one_block_threshold_is_the_same_on_both_sides(built):
    """host.py skips zeroing the accumulator where csrc/softmax.hip runs its single-workgroup form (which overwrites it)."""
    import re
    from cuda_learn_notes_amd import host
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-learn-notes_amd", "csrc", "softmax.hip")).read()
    assert int(re.search(r"SOFTMAX_ONE_BLOCK_MAX = (\d+);", src).group(1)) == host._SOFTMAX_ONE_BLOCK_MAX
