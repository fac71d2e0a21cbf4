"""manifest.py's name -> kernel table is checked against the dispatch code itself: cln_describe() evaluates the
planners of csrc/hgemm.hip and csrc/flash_attn.hip on the host (no GPU) and must return exactly the text the manifest
records for every example shape. (the manifest had drifted from the code.)"""
import pytest


def test_manifest_dispatch_examples_match_the_library(built):
    m = built.manifest
    for name, dims, stages, want in m.DISPATCH_EXAMPLES:
        assert name in m.BY_NAME, name
        got = m.describe(name, dims, stages)
        assert got == want, (name, dims, stages, got)


def test_every_run_time_dispatched_name_describes_itself(built):
    """All FA names and all G6 HGEMM names answer; statically bound names say so (LookupError)."""
    m = built.manifest
    for e in m.ENTRIES:
        if e.sig == "FA" and e.lib != "flash_attn":  # the ck_tile comparison row: a vendor kernel, not one of ours
            with pytest.raises(LookupError):
                m.describe(e.name, (4, 8, 2048, 64), 0)
        elif e.sig == "FA":
            D = 64
            txt = m.describe(e.name, (4, 8, 2048, D), 2)
            assert txt.startswith("fa2_fwd"), (e.name, txt)
        elif e.sig == "G6" and e.lib == "hgemm":
            txt = m.describe(e.name, (4096, 4096, 4096), 2)
            assert txt.startswith(("hgemm_w4", "hgemm_pp", "mfma_ring")), (e.name, txt)
        elif e.sig == "S6":
            txt = m.describe(e.name, (4096, 4096, 4096), 2)
            assert txt.startswith("sgemm_dma<128x128x16"), (e.name, txt)
            with pytest.raises(ValueError):
                m.describe(e.name, (4096, 4096 + 64, 4096), 2)
        elif e.sig == "G3" and e.lib == "hgemm":
            with pytest.raises(LookupError):
                m.describe(e.name, (1024, 1024, 1024), 2)


def test_stages_knob_selects_a_different_flash_attn_kernel(built):
    """stages=1 -> the single-stage form of the stage-2 kernel of the shape (a tile is requested, waited for, then used), stages=2 ->
    prefetching pipeline (reference kStage, share_qkv.cu:843-884); at every head dim, both V layouts."""
    m = built.manifest
    n = "flash_attn_mma_stages_split_q_shared_qkv"
    for shape in ((4, 8, 2048, 64), (1, 48, 8192, 64), (4, 8, 2048, 128), (2, 32, 4096, 256), (1, 2, 256, 96), (4, 8, 1024, 64)):
        one, two = m.describe(n, shape, 1), m.describe(n, shape, 2)
        assert one.replace("load-then-compute", "prefetch") == two + " [single stage: every tile fetch waited for where it is issued]", (shape, one, two)
        assert "stages ignored" not in one + two
    one = m.describe(n + "_swizzle_qkv", (4, 8, 2048, 64), 1)
    assert one.startswith("fa2_fwd_m16x<D=64") and "V^T" in one and "single stage" in one
    # above D = 256 too (reference kStage of the tiling kernels, flash_attn_mma_tiling_qkv.cu:63, :189-223): config C5
    tq = "flash_attn_mma_stages_split_q_tiling_qkv"
    one5, two5 = m.describe(tq, (1, 32, 4096, 512), 1), m.describe(tq, (1, 32, 4096, 512), 2)
    # above D = 256 stages = 1 is the SAME kernel family with every tile fetch waited for where it is issued
    assert one5.startswith("fa2_fwd_pair2<D=512") and "single stage" in one5
    assert two5.startswith("fa2_fwd_pair2<D=512") and "single stage" not in two5 and "stages ignored" not in two5
    for D in (320, 384, 640, 768, 1024):
        for N in (4096, 4160 if D >= 640 else 4224):  # sequence lengths the stage-2 kernel tiles are tiled by stage 1 too
            one, two = m.describe(tq, (1, 16, N, D), 1), m.describe(tq, (1, 16, N, D), 2)
            assert one == two + " [single stage: every tile fetch waited for where it is issued]", (D, N, one, two)
    # the split-KV rung is its own kernel, not an alias of the split-Q dispatcher
    assert m.describe("flash_attn_mma_stages_split_kv", (4, 8, 2048, 64), 2).startswith("fa2_fwd_splitkv")
    with pytest.raises(ValueError):
        m.describe("flash_attn_mma_stages_split_q", (1, 1, 256, 256), 2)  # head dim above this name's limit


def test_hgemm_tile_policy_invariants_over_a_grid_of_shapes(built):
    """csrc/hgemm.hip best_plan, through cln_describe (host only): every M, N, K that are multiples of 64 (K >= 64) get a
    kernel; the tile the text names divides M and N; the one-wave-per-SIMD kernel is only named when K has >= 6 whole
    64-wide tiles (>= 7 when their number is odd) and stages = 2 for the 256x256 form; NN and TN names agree on the tile."""
    import re
    m = built.manifest
    nn = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    tn = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4"
    sizes = [64, 128, 192, 256, 320, 384, 512, 640, 960, 1024, 1280, 1536, 1600, 1920, 2048, 2240, 2560, 3072, 3200, 4096, 4160, 4352, 4800]
    ks = [64, 128, 320, 384, 448, 512, 576, 1024, 4096, 4160, 8192]
    seen = set()
    for M in sizes:
        for N in sizes[::2] + [M]:
            for K in ks:
                for stages in (2, 3):
                    t = m.describe(nn, (M, N, K), stages)
                    fam, bm, bn = re.match(r"(\w+)<(\d+)x(\d+)", t).groups()
                    bm, bn = int(bm), int(bn)
                    assert M % bm == 0 and N % bn == 0, (M, N, K, t)
                    seen.add((fam, bm, bn))
                    if "tail split" in t:  # a few 256 x 256 tiles past whole rounds (csrc/hgemm.hip tail_plan)
                        m_split, rows_b, S, kl = map(int, re.search(r"on rows \[0, (\d+)\) \+ the last (\d+) tile rows as split-K x (\d+) \(K (\d+) per", t).groups())
                        assert (bm, bn) == (256, 256) and m_split + 256 * rows_b == M and kl * S == K and (M // 256) * (N // 256) > 256, (M, N, K, t)
                        seen.add(("tail", bm, bn))
                    elif "split-K" in t:  # few tiles, long K (csrc/hgemm.hip splitk_plan): S whole slices, each a K the peeled structure covers
                        S, kl = map(int, re.search(r"split-K x (\d+) \(K (\d+) per workgroup", t).groups())
                        assert fam == "hgemm_w4" and 2 <= S <= 32 and kl * S == K and K >= 4096 and M * N <= 2048 * 2048, (M, N, K, t)
                        assert kl % 64 == 0 and kl // 64 >= (7 if (kl // 64) & 1 else 6), (M, N, K, t)
                        assert S * M * N * 4 <= 256 << 20 and "stages ignored" in t, (M, N, K, t)
                        seen.add(("split-K", bm, bn))
                    elif fam == "hgemm_w4":
                        nt = K // 64
                        assert K % 64 == 0 and nt >= (7 if nt & 1 else 6), (M, N, K, t)
                        assert not (bm == 256 and bn == 256 and stages != 2), (M, N, K, stages, t)
                        if (bm, bn) != (256, 256):  # the other tile forms have one pipeline and say so
                            assert ("stages ignored" in t) == (stages != 2), (M, N, K, stages, t)
                    assert m.stages_honoured(nn, (M, N, K), stages) == ("stages ignored" not in t), (M, N, K, stages, t)  # the status form of the same fact
                    t2 = m.describe(tn, (M, N, K), stages)
                    assert t2.replace(",TN>", ",NN>") == t and ",TN>" in t2, (t, t2)
    # the policy actually uses its repertoire on this grid
    for want in (("hgemm_w4", 256, 256), ("hgemm_w4", 160, 160), ("hgemm_w4", 192, 192), ("hgemm_w4", 128, 256),
                 ("mfma_ring", 64, 64), ("hgemm_pp", 256, 256), ("split-K", 128, 256), ("split-K", 160, 160), ("split-K", 192, 192), ("split-K", 256, 256), ("tail", 256, 256)):
        assert want in seen, (want, sorted(seen))
    # a shape no tile divides is refused, not mis-tiled
    with pytest.raises(ValueError):
        m.describe(nn, (4096, 4096, 4096 + 16), 2)
    with pytest.raises(ValueError):
        m.describe(nn, (4096 + 32, 4096, 4096), 2)


def test_flash_attn_planner_invariants_over_a_grid_of_shapes(built):
    """csrc/flash_attn.hip fa2_plan through cln_describe: which kernel family serves which shape, and that the family's
    row / key granularity divides N (a launcher would otherwise refuse at run time what describe promised)."""
    m = built.manifest
    tq, sq = "flash_attn_mma_stages_split_q_tiling_qkv", "flash_attn_mma_stages_split_q_shared_qkv"
    rows_per_wg = {"fa2_fwd_m16": 256, "fa2_fwd_pair2": 128, "fa2_fwd_m16x": 256, "fa2_fwd_m16x64r": 512, "fa2_fwd_dsplit": 128, "fa2_fwd_dw4": 64, "fa2_fwd_v2": 64}
    fam_seen = set()
    for D in (32, 64, 96, 128, 256, 320, 384, 512, 640, 768, 1024):
        for (B, H) in ((1, 1), (1, 8), (4, 8), (1, 48), (2, 96), (1, 256)):
            for N in (128, 256, 512, 1024, 2048, 4096, 8192):
                try:
                    t = m.describe(tq, (B, H, N, D), 2)
                except ValueError:  # a shape the rung refuses (N below a kernel's granularity): refusing is allowed, mis-planning is not
                    continue
                fam = t.split("<")[0]
                fam_seen.add((fam, D))
                rows = rows_per_wg[fam]  # (fa2_fwd_pair2, D = 512: 4 pairs of waves x 32 rows)
                assert N % rows == 0 or fam == "fa2_fwd_v2", (B, H, N, D, t)
                wgs256 = B * H * (N // 256) if N % 256 == 0 else 0
                if D == 64 and N == 256:  # one row block per head, two key tiles: the 4-wave v2 kernel at every grid size
                    assert fam == "fa2_fwd_v2" and "NW=4" in t, (B, H, N, D, t)
                elif D in (64, 128, 256) and wgs256 >= (129 if D == 64 else 1) and not (D == 64 and t.startswith("fa2_fwd_m16x64r")):
                    assert fam == ("fa2_fwd_m16" if D == 256 else "fa2_fwd_m16x") and "16x16x32" in t, (B, H, N, D, t)
                if fam == "fa2_fwd_v2":  # 4-wave workgroups whenever N allows; 8 only at D = 32 / 96 on large grids; never 2 when 4 divide N
                    nw = int(t.split("NW=")[1].split(",")[0])
                    assert nw == 4 or (nw == 8 and D in (32, 96) and wgs256 > (256 if D == 32 else 128)) or (nw == 2 and N % 128 != 0), (B, H, N, D, t)
                if D in (320, 384):
                    assert fam == "fa2_fwd_pair2" and "LDS geometry of D=512" in t, t
                if D == 512:
                    assert fam == "fa2_fwd_pair2" and "rows split for QK^T and the softmax, d for PV" in t, t
                if D in (640, 768, 1024):
                    assert fam == "fa2_fwd_dw4" and "one per SIMD" in t, t
                if D <= 256:  # the shared-QKV name (max head dim 256) plans the same kernel
                    assert m.describe(sq, (B, H, N, D), 2) == t
    for want in (("fa2_fwd_m16x", 64), ("fa2_fwd_m16x", 128), ("fa2_fwd_m16", 256), ("fa2_fwd_m16x64r", 64), ("fa2_fwd_v2", 32),
                 ("fa2_fwd_pair2", 384), ("fa2_fwd_pair2", 512), ("fa2_fwd_dw4", 1024), ("fa2_fwd_dw4", 640)):
        assert want in fam_seen, (want, sorted(fam_seen))
    with pytest.raises(ValueError):  # "headdim not support!" of the shared-QKV rung (MAX_HEADDIM_CFG: 256)
        m.describe(sq, (1, 32, 4096, 512), 2)
    with pytest.raises(ValueError):
        m.describe(tq, (1, 8, 100, 64), 2)


def test_stages_honoured_status(built):
    """cln_stages_honoured: a `stages` value the plan cannot act on is reported as a status, not only as text."""
    m = built.manifest
    nn = "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem"
    assert m.stages_honoured(nn, (4096, 4096, 4096), 2) and m.stages_honoured(nn, (4096, 4096, 4096), 3)  # 256 x 256: w4 / ring of slots
    assert m.stages_honoured(nn, (5120, 5120, 5120), 2) and not m.stages_honoured(nn, (5120, 5120, 5120), 4)  # 160 x 160: one pipeline
    assert not m.stages_honoured(nn, (1024, 1024, 16384), 2)  # split-K
    fa = "flash_attn_mma_stages_split_q_tiling_qkv"
    assert m.stages_honoured(fa, (1, 16, 4096, 768), 1) and m.stages_honoured(fa, (1, 16, 4096, 768), 2)
    with pytest.raises(ValueError):
        m.stages_honoured(nn, (100, 100, 100), 2)
    with pytest.raises(LookupError):
        m.stages_honoured("hgemm_naive_f16", (256, 256, 256), 2)
