#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_hgemm.py -m gpu -q > $OUT/c6_hgemm_tests.log 2>&1; echo "hgemm tests rc=$?"; tail -4 $OUT/c6_hgemm_tests.log
timeout 300 ./cuda-learn-notes_amd/harness/hgemm_bench 200 > $OUT/c6_hgemm_bench_cpp.log 2>&1; echo "harness rc=$?"
cat $OUT/c6_hgemm_bench_cpp.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c6_prof_harness -o h -- $GRAFT_REPO_ROOT/cuda-learn-notes_amd/harness/hgemm_bench 100 1024 2048 3072 > $OUT/c6_prof_harness.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$OUT/c6_prof_harness/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-90s calls %6s avg %9.0f ns min %9s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
