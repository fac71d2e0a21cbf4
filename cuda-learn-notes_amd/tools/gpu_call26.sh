#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 150 python -m pytest tests/test_gpu_flash_attn.py -m gpu -x -q --timeout 60 -k "16x16x32 or golden or 512" > $OUT/c26_tests.log 2>&1; echo rc=$?; tail -12 $OUT/c26_tests.log | cut -c1-300
