#!/bin/bash
# odd K tile counts on hgemm_w4: hgemm GPU tests + shapes probe at 4160 / 4800 (K / 64 odd)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_hgemm.py -m gpu -x -q > $OUT/c16_tests.log 2>&1; tail -3 $OUT/c16_tests.log
W4_SHAPES=2,5 timeout 300 python cuda-learn-notes_amd/tools/hg_w4_shapes_probe.py 4160 4800 > $OUT/w4_shapes5.log 2>&1; grep -v amdgpu.ids $OUT/w4_shapes5.log
