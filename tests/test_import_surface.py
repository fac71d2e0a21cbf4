"""CPU: the package imports under its Python name and `import toy_hgemm` -- the module the reference's HGEMM scripts import
(kernels/hgemm/tools/utils.py:116-132) -- carries exactly the names of the reference's pybind module."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CC = "/root/reference/kernels/hgemm/pybind/hgemm.cc"


def test_plain_imports_from_the_repository_root():
    code = ("import cuda_learn_notes_amd as p, toy_hgemm as t; from cuda_learn_notes_amd import bench_utils, manifest;"
            "import __graft_entry__ as g; assert g.load_package() is p;"
            "f = t.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem; print(len(t.__all__), f.__name__, bench_utils.PEAK_FP16_MFMA_TFLOPS)")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    n, name, peak = r.stdout.split()
    assert int(n) == 38 and name == "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem" and float(peak) == 2500.0


def test_toy_hgemm_names_are_the_reference_module(built):
    import toy_hgemm
    names = set(toy_hgemm.__all__)
    assert names == {e.name for e in built.manifest.ENTRIES if e.lib in ("hgemm", "hgemm_vendor")}
    if os.path.exists(REF_CC):  # this container only: the reference's TORCH_BINDING_COMMON_EXTENSION list
        ref = set(re.findall(r"TORCH_BINDING_COMMON_EXTENSION\((\w+)\)", open(REF_CC).read())) - {"func"}  # (the macro definition itself)
        assert ref == names, sorted(ref ^ names)
    for n in sorted(names):
        assert callable(getattr(toy_hgemm, n)), n
    try:
        toy_hgemm.not_a_kernel
        raise AssertionError("unknown attribute must raise")
    except AttributeError:
        pass
