"""CPU: every attention / one-wave-per-SIMD HGEMM kernel that is linked into the PRODUCT library is reachable from the
dispatch code, and everything the dispatch code can name is linked in (round 2 shipped two attention
instantiations -- fa2_fwd_w4_kernel<64,8>, <128,0> -- that no plan could select).

Kernel list: the host-side kernel handles of libcln_amd.so (`nm`: one weak object per __global__ instantiation, named as the
kernel). Reachable set: cln_describe() evaluated over a grid of (name, shape, stages) -- the same planner code the launch
path runs (csrc/flash_attn.hip fa2_plan, csrc/hgemm.hip best_plan)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_handles(so):
    nm, filt = shutil.which("nm"), shutil.which("c++filt")
    if not nm or not filt:
        pytest.skip("binutils nm / c++filt not available")
    out = subprocess.run([nm, so], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "VvWwDd" and "_kernel" in ln and "__device_stub__" not in ln]  # template kernels: weak objects, plain kernels: data objects
    dem = subprocess.run([filt], input="\n".join(n.replace("DF16_", "Dh") for n in names), capture_output=True, text=True, check=True).stdout
    res = []
    for d in dem.splitlines():
        m = re.match(r"(?:void )?((?:\w+::)*\w+_kernel)(?:<(.*?)>)?\(", d)
        if m:
            res.append((m.group(1), [a.strip() for a in (m.group(2) or "").split(",")]))
    return res


def test_attention_kernels_in_the_product_library_are_exactly_the_plannable_ones(built):
    from cuda_learn_notes_amd import _loader
    m = built.manifest
    linked = set()
    for fam, a in kernel_handles(_loader.so_path("libcln_amd.so")):
        if not fam.startswith(("fa2::", "fa::")):
            continue
        f = fam.split("::")[1]
        if f == "fa2_fwd_v2_kernel":  # 4th argument: option bits; 262144 (OPT_1STAGE) = the single-stage form `stages = 1` selects
            # 16384 (OPT_PRE) = fp16 pre-scaled Q (D <= 128; absent: scores scaled in fp32 -- D = 256, and the *_acc_f32 names at D <= 128)
            linked.add(("fa2_fwd_v2", int(a[0]), int(a[1]), a[2] == "true", bool(int(a[3]) & 262144), int(a[0]) <= 128 and not int(a[3]) & 16384))
        elif f == "fa2_fwd_dsplit_kernel":
            linked.add(("fa2_fwd_dsplit", int(a[5]) or int(a[0])))
        elif f == "fa2_fwd_m16_pair_kernel":
            # fragment depth 2, scores scaled in fp32, no debug bits (262144 = the single-stage form `stages = 1` selects: one burst per tile)
            # (PAIR = true, the round-3 .. 5 kernel of D = 512, is a probe-library kernel since round 6: fa2_fwd_pair2 runs that head dim)
            assert a[:3] == ["2", "false", "false"] and a[3] in ("0", "262144"), a
            linked.add(("fa2_fwd_m16", 256, a[3] == "262144"))
        elif f == "fa2_fwd_pair2_kernel":  # <K fragments in flight, V fragments in flight, option bits>: D = 512, round 6
            # option bits: 2 = swizzled fragment bases pinned in registers, two tiles per loop iteration (always); 1 = the single-stage form
            # 4th argument: the real head dim on the D = 512 LDS geometry (0 = 512)
            assert a[:2] == ["4", "2"] and a[2] in ("2", "3") and a[3] in ("0", "320", "384"), a
            linked.add(("fa2_fwd_pair2", int(a[3]) or 512, a[2] == "3"))
        elif f == "fa2_fwd_m16x_kernel":  # 6th argument: option bits (32768 = single-stage form); 7th: V given transposed ([B,H,D,N], the *_swizzle_qkv names)
            # the shipped options: phase-A priority + split prologue (1 << 18: fp32-scaled scores; 1 << 19: row sums on the matrix pipe, with them at 32 rows per wave)
            assert int(a[5]) & ~(32768 | (3 << 16) | (1 << 18) | (1 << 19)) == 5, a
            assert bool(int(a[5]) & (1 << 19)) == (bool(int(a[5]) & (1 << 18)) and a[1] == "32"), a
            linked.add(("fa2_fwd_m16x64r" if a[1] == "64" else "fa2_fwd_m16x", int(a[0]), a[6] == "true", bool(int(a[5]) & 32768), bool(int(a[5]) & (1 << 18))))
        elif f == "fa2_fwd_splitkv_kernel":
            linked.add((f[:-len("_kernel")], int(a[0])))
        elif f == "fa2_fwd_dw4_kernel":  # <D, option bits (1 = the single-stage form), K / V fragments in flight>: round 5, head dims 640 / 768 / 1024
            # 112 = last MFMA group carried across the barrier + M0-walking tile requests + softmax in four sections
            # (D = 640: the carry alone -- 781 vs 744 TF with all three, profiles/r05_fa_dw4_probe_options_fixed.log)
            # + 128: two tiles per loop iteration (compile-time ring-slot parity)
            assert a[1] in (("144", "145") if a[0] == "640" else ("240", "241")) and a[2:] == ["2", "2"], a
            linked.add(("fa2_fwd_dw4", int(a[0]), a[1] in ("241", "145")))
        else:
            raise AssertionError("attention kernel family the planner does not know: %s<%s>" % (fam, ", ".join(a)))
    plannable = set()
    names = [("flash_attn_mma_stages_split_kv", False), ("flash_attn_mma_stages_split_q_shared_qkv", False),
             ("flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv", True), ("flash_attn_mma_stages_split_q_tiling_qkv", False),
             ("flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv", True), ("flash_attn_mma_stages_split_q_shared_qkv_acc_f32", False),
             ("flash_attn_mma_stages_split_q_tiling_qkv_acc_f32", False)]
    for name, vt in names:
        for D in (32, 64, 96, 128, 256, 320, 384, 512, 640, 768, 1024):
            for (B, H) in ((1, 1), (1, 8), (4, 8), (1, 48), (2, 96)):
                for N in (64, 128, 192, 256, 512, 1024, 2048, 4096, 8192):
                    for stages in (1, 2):
                        try:
                            t = m.describe(name, (B, H, N, D), stages)
                        except ValueError:
                            continue
                        fam = t.split("<")[0]
                        d = int(re.search(r"<D=(\d+)", t).group(1))
                        one = "single stage" in t
                        assert one == (stages == 1) or fam == "fa2_fwd_splitkv", t  # (the split-KV rung says load-then-compute in its template text)
                        f32s = "fp32-scaled scores" in t
                        assert f32s == ("_acc_f32" in name and d <= 128), t
                        if fam == "fa2_fwd_v2":
                            plannable.add((fam, d, int(re.search(r"NW=(\d+)", t).group(1)), vt, one, f32s))
                        elif fam in ("fa2_fwd_m16x", "fa2_fwd_m16x64r"):
                            assert ("V^T" in t) == vt, t
                            plannable.add((fam, d, vt, one, f32s))
                        elif fam in ("fa2_fwd_m16", "fa2_fwd_pair2", "fa2_fwd_dw4"):
                            assert not vt, t
                            plannable.add((fam, d, one))
                        else:
                            assert not vt, t
                            plannable.add((fam, d))
    assert linked - plannable == set(), sorted(linked - plannable)   # nothing dead in the product library
    assert plannable - linked == set(), sorted(plannable - linked)   # nothing the planner names is missing


def test_one_wave_per_simd_hgemm_instantiations_are_exactly_the_plannable_ones(built):
    from cuda_learn_notes_amd import _loader
    m = built.manifest
    linked = set()
    linked_s = set()
    linked_k = set()  # split-K forms (csrc/hgemm_splitk.cuh)
    linked_k6 = set()  # their one-launch twins
    fams = set()
    for fam, a in kernel_handles(_loader.so_path("libcln_amd.so")):
        if fam.startswith("hgemm::"):
            fams.add(fam.split("::")[1])
        if fam == "hgemm::hgemm_w4_kernel":
            # LDS epilogue with non-temporal C stores (3) or the split-K partial store (5), the production schedule, no ablation
            # (6, round 5: the split-K form whose last-arriving workgroup reduces in the same launch -- instantiated for the same shapes as 5)
            assert a[1] in ("3", "5", "6") and a[2:4] == ["26", "0"], a
            if a[1] != "6":
                (linked_k if a[1] == "5" else linked).add((int(a[0]), int(a[4]), int(a[5]), a[6] == "true"))
            else:
                linked_k6.add((int(a[0]), int(a[4]), int(a[5]), a[6] == "true"))
        if fam == "hgemm::hgemm_w4s_kernel":  # <layout, ring depth, epilogue>: stages 3 / 4 / 5 of the 256x256 names (2 is the probe library's)
            assert a[2] == "3" and a[1] in ("3", "4", "5"), a
            linked_s.add((int(a[0]), int(a[1])))
    assert linked_s == {(l, s) for l in (0, 1) for s in (3, 4, 5)}, sorted(linked_s)
    for name, layout in (("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", "NN"), ("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4", "TN"),
                         ("hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem", "NN")):
        for st in (3, 4, 5):
            t = m.describe(name, (4096, 4096, 4096), st)
            assert t.startswith("hgemm_w4s<256x256,ring of %d x 32-deep" % st) and t.endswith(layout + ">"), t
            assert not m.describe(name, (4096, 4096, 32 * 2 * st - 64), st).startswith("hgemm_w4s"), st  # fewer than 2 S slots: another kernel
    assert fams == {"hgemm_w4_kernel", "hgemm_w4s_kernel", "hgemm_splitk_reduce_kernel", "hgemm_pp_kernel", "hgemm_pp32_kernel", "hgemm_ring_kernel", "hgemm_1stage_kernel", "hgemm_mfma_naive_kernel",
                    "hgemm_valu_tile_kernel", "hgemm_naive_f16_kernel", "hgemm_sliced_k_f16_kernel"}, sorted(fams)
    plannable = set()
    plannable_k = set()
    rows = [("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", 0), ("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4", 1),
            ("hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem", 0), ("hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem", 0),
            ("hgemm_mma_stages_block_swizzle_tn_cute", 1)]
    sizes = [256, 320, 384, 512, 768, 960, 1280, 1920, 2304, 2560, 2816, 3072, 3200, 4096, 4608, 4800, 6144]
    for name, layout in rows:
        for M in sizes:
            for N in sizes:
                for K in (384, 448, 512, 4096, 4160, 4480, 8192, 8960, 12288):
                    if K > 4160 and M * N > 2048 * 2048:
                        continue
                    try:
                        t = m.describe(name, (M, N, K), 2)
                    except ValueError:
                        continue
                    mm = re.match(r"hgemm_w4<(\d+)x(\d+)x64", t)
                    sk = re.search(r"split-K x \d+ \(K (\d+) per workgroup", t)
                    if mm and sk:
                        plannable_k.add((layout, int(mm.group(1)), int(mm.group(2)), bool((int(sk.group(1)) // 64) & 1)))
                    elif mm:
                        plannable.add((layout, int(mm.group(1)), int(mm.group(2)), bool((K // 64) & 1)))
    assert linked - plannable == set(), sorted(linked - plannable)
    assert plannable - linked == set(), sorted(plannable - linked)
    assert linked_k - plannable_k == set(), sorted(linked_k - plannable_k)
    assert plannable_k - linked_k == set(), sorted(plannable_k - linked_k)
    assert linked_k6 == linked_k, sorted(linked_k6 ^ linked_k)  # the one-launch form exists for exactly the shapes of the two-launch form
