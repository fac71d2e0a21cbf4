#!/bin/bash
# D = 256 on the 16x16x32 kernel in production: attention tests (stop at the first failure), then dispatcher vs the former kernel (220)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_flash_attn.py -m gpu -x -q --timeout 60 > $OUT/c29_tests.log 2>&1 || { echo "tests failed"; tail -15 $OUT/c29_tests.log | cut -c1-300; exit 8; }
tail -2 $OUT/c29_tests.log
FA_PP2=220 timeout 60 python cuda-learn-notes_amd/tools/fa_w4_probe.py 600 "4,8,2048,256;2,32,4096,256" > $OUT/fa_d256_dispatch.log 2>&1
grep -v amdgpu.ids $OUT/fa_d256_dispatch.log | grep "CHK\|^FA" | grep -v "w4 600"
