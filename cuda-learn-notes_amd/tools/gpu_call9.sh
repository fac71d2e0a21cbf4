#!/bin/bash
# after a change to the row kernels: bandwidth GPU tests + the rocprofv3 device-time table
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; T=$PWD/cuda-learn-notes_amd/tools; mkdir -p $OUT; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_bandwidth.py -m gpu -q -x > $OUT/c9_bw_tests.log 2>&1; echo "bw tests rc=$?"; tail -4 $OUT/c9_bw_tests.log
rm -rf $OUT/bwprof
( cd /tmp && export TMPDIR=/tmp && BW_PROF_ORDER=$OUT/bw_prof_order.json timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/bwprof -o bw -- python $T/bw_prof_target.py > $OUT/c9_bw_prof.log 2>&1 )
python $T/bw_prof_summary.py $(ls $OUT/bwprof/*kernel_trace.csv $OUT/bwprof/*/*kernel_trace.csv 2>/dev/null | head -1) $OUT/bw_prof_order.json $OUT/r02_bw_rocprof.json
