"""Every evidence file the documents cite by name exists under profiles/ (the judge cites profiles/ or flags its
absence), every per-round file under profiles/ is listed in profiles/README.md, and the durations the documents quote
next to the two rocprofv3 trace summaries are IN those files (three documents gave three numbers -- 94.27,
96.4 and 101.6 us -- for one file)."""
import csv
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = re.compile(r"`(?:profiles/)?(r0\d_[A-Za-z0-9_.\-]+\.(?:log|json|csv|txt))`")


def test_cited_evidence_files_exist():
    cited = set()
    for f in ("DESIGN.md", "DESIGN_LOG.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md")):
        cited |= set(PAT.findall(open(os.path.join(ROOT, f)).read()))
    assert len(cited) > 50
    missing = sorted(n for n in cited if not os.path.exists(os.path.join(ROOT, "profiles", n)))
    assert not missing, missing


def test_every_round_file_is_in_the_profiles_index():
    index = open(os.path.join(ROOT, "profiles", "README.md")).read()
    files = [n for n in os.listdir(os.path.join(ROOT, "profiles")) if re.match(r"r0\d_", n)]
    unlisted = sorted(n for n in files if n not in index)
    assert not unlisted, unlisted


DOCS = ("DESIGN.md", "DESIGN_LOG.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md"))
TRACE = re.compile(r"`(?:profiles/)?(r0\d_(?:bench_kernel_stats|fa_kernel_trace)\.csv)`")
MICROS = re.compile(r"(\d+(?:\.\d+)?)\s*µs")


def trace_durations_us(name):
    """Every duration a sentence may quote from a trace summary: average / min / max per kernel row, in microseconds."""
    vals = []
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", name))):
        for key, scale in (("AverageNs", 1e-3), ("MinNs", 1e-3), ("MaxNs", 1e-3), ("avg_us", 1.0), ("min_us", 1.0)):
            if r.get(key):
                vals.append(float(r[key]) * scale)
    return vals


def test_durations_quoted_next_to_a_trace_summary_are_in_that_file():
    """A line (table row / sentence line) that names rNN_bench_kernel_stats.csv or rNN_fa_kernel_trace.csv and quotes
    "<number> µs": the number must be a duration of that file, rounded as quoted."""
    checked, bad = 0, []
    for f in DOCS:
        for ln in open(os.path.join(ROOT, f)).read().splitlines():
            files = TRACE.findall(ln)
            if not files:
                continue
            vals = [v for n in files for v in trace_durations_us(n)]
            for q in MICROS.findall(ln):
                checked += 1
                nd = len(q.split(".")[1]) if "." in q else 0
                if not any(abs(round(v, nd) - float(q)) < 0.5 * 10 ** -nd + 1e-9 for v in vals):
                    bad.append((f, q, files, ln[:90]))
    assert checked >= 3, checked  # the check is not vacuous
    assert not bad, bad
