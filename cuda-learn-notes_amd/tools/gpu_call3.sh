#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_flash_attn.py tests/test_scripts.py tests/test_gpu_bandwidth.py -m gpu -q -k "pre_scaled" > $OUT/c4_tests.log 2>&1; echo "tests rc=$?"
tail -8 $OUT/c4_tests.log
RB_FEW=1 timeout 420 python cuda-learn-notes_amd/tools/fa_rb_probe.py > $OUT/c4_fa_probe.log 2>&1; echo "probe rc=$?"
grep "^FA\|BAD\|ERR" $OUT/c4_fa_probe.log | cut -c1-150
