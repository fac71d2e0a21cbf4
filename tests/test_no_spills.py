"""Every kernel a C-ABI name can dispatch must keep its working set in registers: `.vgpr_spill_count 0`,
`.sgpr_spill_count 0` and no scratch (`.private_segment_fixed_size 0`), read from the gfx950 assembly hipcc emits for
the production sources (no GPU needed: hipcc cross-compiles). The reference's fine-grained-tiling rungs are O(1)-SRAM
with no local memory (flash_attn_mma_tiling_qkv.cu:70); round 1 shipped a C5 kernel that spilled 15 registers.

Probe-only instantiations that live in the product objects for the test-only hook (ring_exact with the T256W4 tile)
are listed explicitly -- nothing else may spill."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-learn-notes_amd", "tools"))

# 256x256 tile on 4 waves x 128x128 wave tiles: reachable only through cln_hgemm_variant (libcln_amd_probe.so);
# no reference name dispatches it (csrc/hgemm.hip best_tile / the CLN_G6 table)
PROBE_ONLY = ("hgemm::Cfg<256, 256, 32, 2, 2,", "hgemm::Cfg<256, 256, 64, 2, 2,")

SOURCES = ["flash_attn.hip", "flash_attn_m16x.hip", "hgemm.hip", "hgemm_ring_nn.hip", "hgemm_ring_tn.hip", "sgemm.hip", "softmax.hip", "norm.hip",
           "reduce.hip", "elementwise.hip", "rope.hip", "activation.hip", "blas1.hip", "indexing.hip"]


@pytest.mark.parametrize("src", SOURCES)
def test_no_register_spills(src, tmp_path):
    import kernel_resources as kr
    kernels, _ = kr.report(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", src), keep=str(tmp_path))
    assert kernels, src
    bad = [(k["demangled"][:100], k["spill"], k["sgpr_spill"], k["scratch"]) for k in kernels
           if (k["spill"] or k["sgpr_spill"] or k["scratch"]) and not any(p in k["demangled"] for p in PROBE_ONLY)]
    assert not bad, bad


def test_production_attention_instantiations_are_in_the_report(tmp_path):
    """The kernels behind BASELINE configs C4 / C5 are among the checked ones (guards against the check going vacuous
    when an instantiation is renamed)."""
    import kernel_resources as kr
    kernels, _ = kr.report(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", "flash_attn.hip"), keep=str(tmp_path))
    kernels_x, _ = kr.report(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", "flash_attn_m16x.hip"), keep=str(tmp_path))
    names = [k["demangled"] for k in kernels + kernels_x]
    # the product unit holds the three dispatched forms, each for V as [B,H,N,D] and as [B,H,D,N], each as the stage-2 pipeline and in its
    # single-stage form (`stages = 1`); the [B,H,N,D] ones also with the scores scaled in fp32 (the *_acc_f32 names)
    assert len(kernels_x) == 18, [k["demangled"] for k in kernels_x]
    for want in ("fa2_fwd_pair2_kernel<4, 2, 2, 384>", "fa2_fwd_m16x_kernel<64, 32, 128, 8, 4, 5, false>", "fa2_fwd_m16x_kernel<128, 32, 128, 4, 4, 5, false>", "fa2_fwd_m16x_kernel<128, 32, 128, 4, 4, 5, true>", "fa2_fwd_m16_pair_kernel<2, false, false, 0>", "fa2_fwd_pair2_kernel<4, 2, 2, 0>",
                 "fa2_fwd_v2_kernel<128, 2, true", "fa2_fwd_splitkv_kernel<64, true>", "fa2_fwd_dw4_kernel<1024, 240, 2, 2>", "fa2_fwd_dw4_kernel<640, 145, 2, 2>", "fa2_fwd_m16x_kernel<64, 64, 64, 4, 1, 5, false>"):
        assert any(want in n for n in names), want


def test_inline_asm_mfma_stream_of_the_one_wave_per_simd_hgemm(tmp_path):
    """hgemm_w4_kernel issues its MFMAs from inline asm (AGPR-tied accumulators), which hides them from hipcc's hazard
    pass: the wait states must hold by construction (a zero-fill sunk next to the first MFMA produced NaNs on the GPU
    while this kernel was being written). Checked on the code object, both layouts, 512 registers, no spill."""
    import kernel_resources as kr
    kernels, s = kr.report(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", "hgemm.hip"), keep=str(tmp_path))
    w4 = [k for k in kernels if "hgemm_w4_kernel" in k["demangled"]]
    # 256x256, 192x256, 256x192, 192x192, 128x256, 256x128, 160x160 tiles x NN / TN x even / odd K tile count, plus the split-K forms (EPI 5: partial
    # store instead of the LDS epilogue) of 256x256, 192x256, 192x192, 128x256, 160x160
    # (round 5) ... each split-K form twice: EPI 6 = partials + in-kernel fix-up by the last-arriving workgroup (one launch, up to 4 splits),
    # EPI 5 = partials only, summed by the reduce launch (more splits)
    assert len(w4) == 28 + 20 + 20, [k["demangled"] for k in w4]
    for epi in (5, 6):
        assert sum(("hgemm_w4_kernel<0, %d," % epi) in k["demangled"] or ("hgemm_w4_kernel<1, %d," % epi) in k["demangled"] for k in w4) == 20
    text = open(s).read()
    for k in w4:
        assert k["agpr"] in (256, 192, 144, 128, 100) and k["spill"] == 0 and k["scratch"] == 0, k
        assert kr.asm_mfma_stream_check(text, k["name"]) == [], k["demangled"]
        # the one-launch split-K fix-up reads the other splits' partials with agent-scope loads from inline asm; the wait is a separate asm statement:
        # nothing may touch a destination register in between
        if "hgemm_w4_kernel<0, 6," in k["demangled"] or "hgemm_w4_kernel<1, 6," in k["demangled"]:
            assert kr.asm_inflight_load_hazards(text, k["name"]) == [], k["demangled"]
    # the ring-of-slots form (hgemm_w4s.cuh, stages 3 / 4 / 5 of the 256x256 names): same inline-asm MFMAs, same rules
    w4s = [k for k in kernels if "hgemm_w4s_kernel" in k["demangled"]]
    assert len(w4s) == 6, [k["demangled"] for k in w4s]  # ring depth 3 / 4 / 5 x NN / TN
    for k in w4s:
        assert k["agpr"] == 256 and k["vgpr"] <= 512 and k["spill"] == 0 and k["scratch"] == 0, k
        assert kr.asm_mfma_stream_check(text, k["name"]) == [], k["demangled"]


def test_production_attention_kernels_use_the_16x16x32_matrix_shape(tmp_path):
    """attention sits at the package power cap and v_mfma_f32_16x16x32_f16 is the energy-cheaper shape, so the
    kernels behind D = 64 / 128 / 256 must issue ONLY that MFMA; per KV tile and wave the D = 64 kernel's loop holds 64 of
    them (32 for S^T = K Q^T, 32 for O^T += V^T P^T) and 64 exponentials (32 rows x 128 keys / 64 lanes)."""
    import re
    import kernel_resources as kr
    kernels, s = kr.report(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", "flash_attn.hip"), keep=str(tmp_path))
    kernels_x, sx = kr.report(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", "flash_attn_m16x.hip"), keep=str(tmp_path))
    text = open(s).read() + open(sx).read()
    for want in ("fa2_fwd_m16x_kernel<64, 32, 128, 8, 4, 5, false>", "fa2_fwd_m16x_kernel<128, 32, 128, 4, 4, 5, false>", "fa2_fwd_m16x_kernel<128, 32, 128, 4, 4, 5, true>", "fa2_fwd_m16_pair_kernel<2, false, false, 0>",
                 "fa2_fwd_pair2_kernel<4, 2, 2, 0>"):
        k = [k for k in kernels + kernels_x if want in k["demangled"]]
        assert len(k) == 1, want
        body = text[text.index("\n" + k[0]["name"] + ":"):]
        body = body[:body.index("s_endpgm")]
        shapes = set(re.findall(r"v_mfma_f32_(\w+?)_f16", body))
        assert shapes == {"16x16x32"}, (want, shapes)
        if want.startswith("fa2_fwd_m16x_kernel<64"):
            # one tile body in the KV loop: 32 + 32 MFMAs and 64 exponentials per wave (32 rows x 128 keys / 64 lanes);
            # the cold (rescale) block repeats the 32 optimistic ones of phase A and holds no MFMA. The sum-checked
            # softmax has no row-maximum chain on the hot path: the v_max3 of the kernel sit in the cold block and in the
            # check of the deferred key blocks (4 per block and query block)
            assert body.count("v_mfma_f32_16x16x32_f16") == 64, body.count("v_mfma_f32_16x16x32_f16")
            assert 96 <= body.count("v_exp_f32") <= 100, body.count("v_exp_f32")
            assert "v_pk_add_f32" not in body  # -fno-slp-vectorize on this unit (see flash_attn_m16x.hip)


MFMA_SOURCES = ["flash_attn.hip", "flash_attn_m16x.hip", "hgemm.hip", "hgemm_ring_nn.hip", "hgemm_ring_tn.hip", "sgemm.hip"]


@pytest.mark.parametrize("src", MFMA_SOURCES)
def test_no_mfma_destination_on_its_operand_registers(src, tmp_path):
    """Round 3 root cause of the D = 256 attention kernel's ~1 % wrong launches (and of the D = 512 probe that DESIGN r2 9.1
    listed as "wrong, cause not found"): hipcc puts no early-clobber on an MFMA destination, so the last MFMA that reads a
    dying A / B fragment may WRITE ITS RESULT OVER THAT FRAGMENT (`v_mfma_f32_16x16x32_f16 v[66:69], v[66:69], v[34:37], 0`);
    on MI355X such an instruction intermittently returns wrong values when another wave's MFMAs interleave on the same SIMD
    (profiles/r03_fa_mfma_overlap_bisect.log: the failure rate follows the number of such instructions; with none, every
    timing variant is bit-stable). `cln_mfma_keep` (csrc/common.h) keeps the operands alive across the MFMA; this test keeps
    every code object of the product library free of the pattern."""
    import kernel_resources as kr
    import mfma_overlap_scan as scan
    _, s = kr.report(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", src), keep=str(tmp_path))
    text = open(s).read()
    assert text.count("v_mfma") > 0, src
    bad = scan.scan(text)
    assert not bad, [(n[:60], l) for n, l in bad[:6]]


# kernels of the TEST-ONLY probe library that carry the pattern ON PURPOSE or whose results are garbage by design
PROBE_OVERLAP_ALLOWED = (
    "fa2_fwd_m16_pair_kernelILi2ELb1ELb0ELi32768E", "fa2_fwd_m16_pair_kernelILi2ELb0ELb0ELi32768E",  # DBG 32768 = the round-2 code without cln_mfma_keep (the bisect's positive control)
    "fa2_fwd_dsplit_kernelILi64ELi1ELi2ELi147485E",  # row sums on the matrix pipe (OPT_SUMM probe, profiles/r02_fa_rowsum_on_mfma_probe.log): its ones-operand MFMA
    "hgemm_m32_kernelILi0ELi1E", "hgemm_pp_kernelILi0ELi1E",  # EPI = 1: no-store timing ablations (nothing is written back)
)


def test_built_libraries_have_no_mfma_destination_on_operand_registers(built, tmp_path):
    """the per-source scan above covers the product units only, yet the probe library's measurements and
    bit-repeatability tests are the evidence behind the dispatch choices. This scans the code objects INSIDE the built
    libraries (llvm-objdump of the shipped binaries, no recompilation): the product library must be clean; in the probe library
    only the deliberate positive controls and the no-store ablations may carry the pattern."""
    import mfma_overlap_scan as scan
    from cuda_learn_notes_amd import _loader
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    bad, n = scan.scan_shared_object(_loader.so_path("libcln_amd.so"), str(tmp_path / "product"))
    assert n > 20000, n  # the scan saw the library's MFMAs
    assert not bad, [(k[:70], l) for k, l in bad[:6]]
    bad, n = scan.scan_shared_object(_loader.so_path("libcln_amd_probe.so"), str(tmp_path / "probe"))
    assert n > 50000, n
    stray = [(k[:90], l) for k, l in bad if not any(a in k for a in PROBE_OVERLAP_ALLOWED)]
    assert not stray, stray[:6]


def test_built_product_library_has_no_spilling_kernel_a_name_can_reach(built, tmp_path):
    """The same property read from the SHIPPED binary (kernel metadata of the code objects inside libcln_amd.so, llvm-readelf): no kernel spills
    or uses scratch except the probe-only 256x256-tile-on-4-waves ring instantiations that share an object file with the product rings."""
    import mfma_overlap_scan as scan
    from cuda_learn_notes_amd import _loader
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm-readelf not available")
    n, bad = scan.shared_object_spills(_loader.so_path("libcln_amd.so"), str(tmp_path))
    assert n > 400, n
    stray = [b for b in bad if "CfgILi256ELi256ELi32ELi2ELi2E" not in b[0] and "CfgILi256ELi256ELi64ELi2ELi2E" not in b[0]]
    assert not stray, stray


def test_inline_asm_mfmas_of_the_large_head_dim_attention_kernel_never_read_a_fresh_valu_result(tmp_path):
    """csrc/flash_attn_dw4.cuh issues every MFMA from inline asm (O^T tied to AGPRs, S^T to one VGPR tuple): the two wait states a VALU-written A / B
    operand needs are then the kernel's own business. Round 5 shipped a pad in front of the FIRST MFMA of a group only and hipcc scheduled the register
    copy of the second row block's P fragment between the two MFMAs of the group: one 32 x 32 tile of O wrong on the GPU
    (profiles/r05_fa_dw4_unroll2_debug.log). Operands are now pinned + padded once per phase; this scan of the code object holds every MFMA to it."""
    import kernel_resources as kr
    kernels, s = kr.report(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", "flash_attn.hip"), keep=str(tmp_path))
    text = open(s).read()
    dw4 = [k for k in kernels if "fa2_fwd_dw4_kernel" in k["demangled"]]
    assert len(dw4) == 6, [k["demangled"] for k in dw4]  # D = 640 / 768 / 1024 x stages 2 / 1
    for k in dw4:
        assert k["spill"] == 0 and k["scratch"] == 0 and k["agpr"] in (160, 192, 256), k
        assert kr.asm_mfma_operand_hazards(text, k["name"]) == [], k["demangled"]
