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
        if e.sig == "FA":
            D = 64
            txt = m.describe(e.name, (4, 8, 2048, D), 2)
            assert txt.startswith("fa2_fwd"), (e.name, txt)
        elif e.sig == "G6" and e.lib == "hgemm":
            txt = m.describe(e.name, (4096, 4096, 4096), 2)
            assert txt.startswith(("hgemm_w4", "hgemm_pp", "mfma_ring")), (e.name, txt)
        elif e.sig == "G3" and e.lib == "hgemm":
            with pytest.raises(LookupError):
                m.describe(e.name, (1024, 1024, 1024), 2)


def test_stages_knob_selects_a_different_flash_attn_kernel(built):
    """stages=1 -> load-then-compute, stages=2 -> prefetching pipeline (reference kStage, share_qkv.cu:843-884);
    where one pipeline serves both, the description says so."""
    m = built.manifest
    n = "flash_attn_mma_stages_split_q_shared_qkv"
    one, two = m.describe(n, (4, 8, 2048, 64), 1), m.describe(n, (4, 8, 2048, 64), 2)
    assert "load-then-compute" in one and "load-then-compute" not in two
    assert "stages ignored" in m.describe("flash_attn_mma_stages_split_q_tiling_qkv", (1, 32, 4096, 512), 1)
    # the split-KV rung is its own kernel, not an alias of the split-Q dispatcher
    assert m.describe("flash_attn_mma_stages_split_kv", (4, 8, 2048, 64), 2).startswith("fa2_fwd_splitkv")
    with pytest.raises(ValueError):
        m.describe("flash_attn_mma_stages_split_q", (1, 1, 256, 256), 2)  # head dim above this name's limit
