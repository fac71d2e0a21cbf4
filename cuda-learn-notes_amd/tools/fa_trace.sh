# rocprofv3 kernel-trace durations of the attention kernels at the bench shapes (device time, no Python in the way): ONE
# traced process, every shape pre-warmed >= 0.3 s, 200 timed launches each (fa_trace_target.py / fa_trace_summary.py).
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out; T=$REPO/cuda-learn-notes_amd/tools
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fatrace
FA_TRACE_ORDER=$OUT/fa_trace_order.json rocprofv3 --kernel-trace --output-format csv -d $OUT/fatrace -o t -- python $T/fa_trace_target.py 200 > $OUT/fatrace.log 2>&1
python $T/fa_trace_summary.py $(ls $OUT/fatrace/*kernel_trace.csv $OUT/fatrace/*/*kernel_trace.csv 2>/dev/null | head -1) $OUT/fa_trace_order.json $OUT/${TAG}_fa_kernel_trace.csv
