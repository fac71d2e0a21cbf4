"""The microbenchmarks under tools/ubench/ are evidence generators (their logs are cited from DESIGN.md and indexed in
profiles/README.md): every one of them must still compile for gfx950, and every cited log must name an existing source."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UB = os.path.join(ROOT, "cuda-learn-notes_amd", "tools", "ubench")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def test_every_microbenchmark_compiles_for_gfx950(tmp_path):
    srcs = sorted(f for f in os.listdir(UB) if f.endswith(".hip"))
    assert len(srcs) >= 9, srcs

    def build(src):
        out = str(tmp_path / (src + ".o"))
        r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "--cuda-device-only", "-c", os.path.join(UB, src), "-o", out],
                           capture_output=True, text=True)
        return src, r.returncode, r.stderr[-2000:]

    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(build, srcs))
    bad = [(s, err) for s, rc, err in results if rc != 0]
    assert not bad, bad


def test_index_rows_of_microbenchmark_logs_name_existing_sources():
    text = open(os.path.join(ROOT, "profiles", "README.md")).read()
    cited = set(re.findall(r"tools/ubench/([A-Za-z0-9_]+\.hip)", text))
    assert cited, "no microbenchmark rows in profiles/README.md"
    missing = [c for c in cited if not os.path.exists(os.path.join(UB, c))]
    assert not missing, missing
    # and every microbenchmark source is cited by at least one index row or by DESIGN.md
    design = open(os.path.join(ROOT, "DESIGN.md")).read() + open(os.path.join(ROOT, "DESIGN_LOG.md")).read()
    uncited = [f for f in os.listdir(UB) if f.endswith(".hip") and f not in cited and ("ubench/" + f) not in design]
    assert not uncited, uncited
