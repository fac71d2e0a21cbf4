"""CPU: the C-ABI header is valid C and C++ on its own (no HIP, no torch types), and the stand-alone C++ harness
links against the built libraries -- i.e. the boundary really is `extern "C"` + plain pointers and sizes."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "cln_amd.h")


@pytest.mark.parametrize("lang,cc", [("c", "gcc"), ("c++", "g++")])
def test_header_compiles_standalone(tmp_path, lang, cc):
    if not shutil.which(cc):
        pytest.skip(cc + " not available")
    src = tmp_path / ("t.c" if lang == "c" else "t.cpp")
    src.write_text('#include "cln_amd.h"\n'
                   "int (*probe_hgemm)(const void*, const void*, void*, int, int, int, int, int, int, void*) =\n"
                   "    hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem;\n"
                   "int (*probe_fa)(const void*, const void*, const void*, void*, int, int, int, int, int, void*) =\n"
                   "    flash_attn_mma_stages_split_q_shared_qkv;\n"
                   "int (*probe_hist)(const void*, void*, long long, int, void*) = histogram_i32;\n"
                   "int main(void) { return probe_hgemm && probe_fa && probe_hist ? 0 : 1; }\n")
    r = subprocess.run([cc, "-x", lang, "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.dirname(HDR), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_cpp_harness_links_against_the_libraries(built, tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    lib = os.path.join(ROOT, "cuda-learn-notes_amd", "lib")
    out = tmp_path / "hgemm_bench"
    r = subprocess.run([hipcc, "-O1", "-std=c++17", os.path.join(ROOT, "cuda-learn-notes_amd", "harness", "hgemm_bench.cpp"),
                        "-I", os.path.dirname(HDR), "-L", lib, "-lcln_amd", "-lcln_amd_vendor",
                        "-Wl,-rpath," + lib, "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists()


@pytest.mark.parametrize("script", ["histogram/histogram.py", "rope/rope.py", "sgemv/sgemv.py", "elementwise/elementwise.py"])
def test_bench_scripts_run_without_a_gpu(script):
    """BASELINE config C1: the reference's own CPU-runnable path -- the torch rows of the scripts run on CPU and
    the kernel rows are reported as skipped (never silently replaced by a CPU computation)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-path check")
    p = os.path.join(ROOT, "cuda-learn-notes_amd", "kernels", script)
    r = subprocess.run([sys.executable, p], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "skipped (no GPU" in r.stdout
    assert "_th" in r.stdout
