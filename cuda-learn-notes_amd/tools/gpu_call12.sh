#!/bin/bash
# 160x160 one-wave-per-SIMD tile in the product policy: hgemm GPU tests, shapes probe at the sizes it changes, C++ harness
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_hgemm.py -m gpu -x -q > $OUT/c12_tests.log 2>&1; tail -3 $OUT/c12_tests.log
W4_SHAPES=3,5 timeout 300 python cuda-learn-notes_amd/tools/hg_w4_shapes_probe.py 1920 3200 4480 4800 5120 6400 > $OUT/w4_shapes4.log 2>&1; grep -c OK $OUT/w4_shapes4.log; grep -c BAD $OUT/w4_shapes4.log
timeout 300 cuda-learn-notes_amd/harness/hgemm_bench 100 1024 1920 2048 2560 3072 3200 4096 > $OUT/c12_harness.log 2>&1; tail -30 $OUT/c12_harness.log
