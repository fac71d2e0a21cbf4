#!/bin/bash
# m16 attention kernel as the D = 64 / 128 production path: FA GPU tests; D = 128 with 128-key tiles (543) vs 64-key (540)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_flash_attn.py tests/test_scripts.py -m gpu -x -q > $OUT/c20_tests.log 2>&1; tail -5 $OUT/c20_tests.log
FA_PP2=500,540,543 timeout 300 python cuda-learn-notes_amd/tools/fa_w4_probe.py 600 "4,8,2048,128;2,32,4096,128" > $OUT/fa_m16c.log 2>&1
grep -v amdgpu.ids $OUT/fa_m16c.log | grep "CHK\|^FA\|Error\|error" | grep -v "sdpa\|w4 60"
