#!/bin/bash
# The three evidence pieces of round_evidence.sh that depend on the bandwidth / row kernels only, for a refresh after those kernels change:
# rocprofv3 kernel-trace of the bandwidth families (-> r05_bw_rocprof.json / .txt), every reference-style script on the GPU, every script against its torch row.
TAG=r05; REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out; T=$REPO/cuda-learn-notes_amd/tools
mkdir -p $OUT; cd $REPO; export PYTHONUNBUFFERED=1
rm -rf $OUT/bwprof
( cd /tmp && export TMPDIR=/tmp && BW_PROF_ORDER=$OUT/bw_prof_order.json timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/bwprof -o bw -- python $T/bw_prof_target.py > $OUT/${TAG}_bw_prof.log 2>&1 )
python $T/bw_prof_summary.py $(ls $OUT/bwprof/*kernel_trace.csv $OUT/bwprof/*/*kernel_trace.csv 2>/dev/null | head -1) $OUT/bw_prof_order.json $OUT/${TAG}_bw_rocprof.json > $OUT/${TAG}_bw_rocprof.txt 2>&1; echo "bw rc=$?"
timeout 900 bash $T/run_all_scripts.sh > $OUT/${TAG}_reference_style_scripts_on_gpu.log 2>&1; echo "scripts rc=$?"
timeout 900 python $T/scripts_vs_torch.py 2>&1 | grep "^SVT" > $OUT/${TAG}_scripts_vs_torch.log; echo "scripts vs torch rc=$?"
grep SVTSUM $OUT/${TAG}_scripts_vs_torch.log | head -12
grep rotating $OUT/${TAG}_bw_rocprof.txt
