"""kernels/hgemm/tools/print_lds_swizzle.py (the analogue of the reference's kernels/hgemm/tools/print_swizzle_layout.py) prints the layouts
the kernels use: its formulas are held to the C++ ones by text and to the conflict-freedom the kernels' comments claim by enumeration."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "cuda-learn-notes_amd", "kernels", "hgemm", "tools", "print_lds_swizzle.py")
sys.path.insert(0, os.path.dirname(TOOL))
import print_lds_swizzle as pls  # noqa: E402


def test_formulas_are_the_ones_in_the_kernels():
    src = open(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", "hgemm_mfma.cuh")).read()
    assert "return (row >> 1) & 7;" in src and "return (((t ^ (t >> 1)) & 1) << 1) | (t >> 1);" in src
    assert "return ((krow & 3) << 1) | (((krow >> 3) & 1) << 3);" in src
    assert "return ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1);" in src and "return ((krow >> 3) & 1) << 1;" in src
    dw4 = open(os.path.join(ROOT, "cuda-learn-notes_amd", "csrc", "flash_attn_dw4.cuh")).read()
    assert "(c ^ (r & 15)) << 4" in dw4 and "(c ^ ((r & 3) << 2)) << 4" in dw4
    for r in range(64):
        assert pls.kswz(r, 64) == (r >> 1) & 7 and pls.kswz(r, 32) == [0, 2, 3, 1][(r >> 2) & 3]
        assert pls.attn_k(r) == r & 15 and pls.attn_v(r) == (r & 3) << 2


def test_every_image_is_a_permutation_of_its_row_and_k_reads_are_conflict_free():
    for kind, rows, rb in (("k64", 64, 0), ("k32", 64, 0), ("n256", 64, 0), ("n128", 64, 0), ("n192", 64, 0), ("n160", 64, 0), ("n64", 64, 0),
                           ("attn-k", 16, 2048), ("attn-k", 16, 1536), ("attn-k", 16, 1280), ("attn-v", 16, 2048), ("attn-v", 16, 1280)):
        tab, cpr = pls.layout(kind, rows, rb or 2048)
        for row in tab:
            got = sorted(c for c in row if c is not None)
            assert got == list(range(cpr)), (kind, row)  # the XOR stays inside the row: every logical chunk exactly once
    # one ds_read_b128 pass = 16 rows x 16 bytes of the same logical chunk: all 64 banks (what the K / A / B^T fragment reads rely on)
    for kind, rb in (("k64", 0), ("attn-k", 2048), ("attn-k", 1536), ("attn-k", 1280)):
        for g in (range(0, 16),) + ((range(16, 32),) if kind == "k64" else ()):
            for chunk in range(4):
                assert len(set(pls.banks_of_read(kind, list(g), chunk, rb or 2048))) == 64, (kind, chunk)
    # without the swizzle the same read is an 8-way (128-byte rows: rows alternate between two 4-bank groups) or 16-way (2048-byte rows) conflict
    assert len({((r * 128 + 4 * k) // 4) % 64 for r in range(16) for k in range(4)}) == 8
    assert len({((r * 2048 + 4 * k) // 4) % 64 for r in range(16) for k in range(4)}) == 4


def test_cli_prints_a_table():
    out = subprocess.run([sys.executable, TOOL, "--image", "attn-k", "--row-bytes", "2048", "--rows", "16"], capture_output=True, text=True, check=True).stdout
    assert "16 rows x 128 chunks" in out and len(re.findall(r"conflict-free", out)) == 2
    out = subprocess.run([sys.executable, TOOL, "--image", "k64", "--rows", "32"], capture_output=True, text=True, check=True).stdout
    assert out.count("conflict-free") == 4 and "32 distinct banks" not in out.split("swizzled")[0]
