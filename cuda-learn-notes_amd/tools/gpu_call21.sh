#!/bin/bash
# after the aborted call 20: golden attention test alone first; only if it passes the attention + script tests (per-test
# timeout, stop at the first failure), then the D = 128 128-key-tile probe
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 120 python -m pytest tests/test_gpu_flash_attn.py -m gpu -x -q -k golden_fixture > $OUT/c21_golden.log 2>&1 || { tail -30 $OUT/c21_golden.log; exit 7; }
tail -2 $OUT/c21_golden.log
timeout 500 python -m pytest tests/test_gpu_flash_attn.py tests/test_scripts.py -m gpu -x -q --timeout 90 > $OUT/c21_tests.log 2>&1 || { tail -40 $OUT/c21_tests.log; exit 8; }
tail -3 $OUT/c21_tests.log
FA_PP2=500,540,543 timeout 120 python cuda-learn-notes_amd/tools/fa_w4_probe.py 600 "4,8,2048,128;2,32,4096,128" > $OUT/fa_m16c.log 2>&1
grep -v amdgpu.ids $OUT/fa_m16c.log | grep "CHK\|^FA\|Error\|error" | grep -v "sdpa\|w4 60"
