#!/bin/bash
# fixed-tile rungs on hgemm_w4 + key-split attention probe test + hgemm.py rows
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_hgemm.py tests/test_gpu_flash_attn.py -m gpu -x -q -k "fixed_tile or key_split or policy_sizes or one_wave_per_simd_kernel_on_192" > $OUT/c15_tests.log 2>&1; tail -5 $OUT/c15_tests.log
timeout 300 python cuda-learn-notes_amd/kernels/hgemm/hgemm.py --wmma-all --mma-all --enable-mma-tn --MNK 4096 --iters 20 > $OUT/c15_hgemm_py.log 2>&1; grep TFLOPS $OUT/c15_hgemm_py.log
