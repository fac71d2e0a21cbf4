# rocprofv3 kernel-trace durations of the attention kernels at the bench shapes (device time, no Python in the way).
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out; T=$REPO/cuda-learn-notes_amd/tools
cd /tmp && export TMPDIR=/tmp
echo "shape,kernel,calls,avg_us,tflops_4bhn2d" > $OUT/${TAG}_fa_kernel_trace.csv
for cfg in "4 8 2048 64" "4 8 2048 128" "1 48 8192 64" "2 32 4096 128" "2 32 4096 256" "1 32 4096 512" "1 16 4096 768" "1 16 4096 1024"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fatrace_$tag -o t -- python $T/prof_target.py fa $cfg 2 20 > $OUT/fatrace_$tag.log 2>&1
  python - "$cfg" $OUT/fatrace_$tag/t_kernel_stats.csv >> $OUT/${TAG}_fa_kernel_trace.csv <<'PY'
import csv, sys
B, H, N, D = map(int, sys.argv[1].split())
for r in csv.DictReader(open(sys.argv[2])):
    if "fa2_fwd" in r["Name"] or "fa2" in r["Name"]:
        us = float(r["AverageNs"]) / 1e3
        print("%dx%dx%dx%d,%s,%s,%.2f,%.1f" % (B, H, N, D, r["Name"][:70].replace(",", ";"), r["Calls"], us, 4.0 * B * H * N * N * D / us * 1e-6))
PY
done
cat $OUT/${TAG}_fa_kernel_trace.csv
