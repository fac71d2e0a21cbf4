#!/bin/bash
# rocprofv3 kernel-trace stats of the C++ harness at the small / mid sizes (which kernel runs behind each name, device time)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/c11_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c11_prof -o h -- $GRAFT_REPO_ROOT/cuda-learn-notes_amd/harness/hgemm_bench 100 1024 2048 2560 3072 > $OUT/c11_prof.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$OUT/c11_prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    with open("$OUT/r02_hgemm_harness_kernel_stats.csv","w") as o:
        w=csv.writer(o); w.writerow(["Name(140)","Calls","AverageNs","MinNs","MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:140], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
            if int(r["Calls"])>50: print("%-120s calls %6s avg %9.0f ns" % (r["Name"][:120], r["Calls"], float(r["AverageNs"])))
PY
