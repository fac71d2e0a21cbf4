#!/bin/bash
# after the 64x64 small-problem tile: hgemm tests, small-size probe log, C++ harness
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_hgemm.py -m gpu -q -x > $OUT/c10_hgemm_tests.log 2>&1; echo "hgemm tests rc=$?"; tail -4 $OUT/c10_hgemm_tests.log
timeout 300 python cuda-learn-notes_amd/tools/hg_small_probe.py 512 768 1024 1280 1536 1792 2048 2>&1 | grep HS > $OUT/r02_hgemm_small_probe.log; grep "top rung stages=2\|rocblas" $OUT/r02_hgemm_small_probe.log
timeout 300 ./cuda-learn-notes_amd/harness/hgemm_bench 200 > $OUT/r02_hgemm_bench_cpp.log 2>&1; echo "harness rc=$?"
grep -v "max |err|" $OUT/r02_hgemm_bench_cpp.log | head -14
