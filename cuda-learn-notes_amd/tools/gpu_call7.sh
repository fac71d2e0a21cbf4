#!/bin/bash
# rocBLAS kernel names + durations at 4096 / 8192 (which Tensile solution wins TN), from the C++ harness
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c7_prof_harness -o h -- $GRAFT_REPO_ROOT/cuda-learn-notes_amd/harness/hgemm_bench 100 4096 8192 > $OUT/c7_prof_harness.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$OUT/c7_prof_harness/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-230s calls %6s avg %9.0f ns min %9s" % (r["Name"][:230], r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
tail -20 $OUT/c7_prof_harness.log
