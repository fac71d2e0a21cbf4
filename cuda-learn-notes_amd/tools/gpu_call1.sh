#!/bin/bash
# round-2 GPU call 1: new FA kernels first (bounded by timeouts), then the whole GPU suite, then the bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_flash_attn.py -m gpu -q -k "split_kv or stages_one or register_blocked" > $OUT/c1_new_fa_tests.log 2>&1; echo "new FA tests rc=$?"
tail -5 $OUT/c1_new_fa_tests.log
timeout 420 python cuda-learn-notes_amd/tools/fa_rb_probe.py > $OUT/c1_fa_rb_probe.log 2>&1; echo "rb probe rc=$?"
grep "^FA\|BAD\|ERR" $OUT/c1_fa_rb_probe.log | cut -c1-150
timeout 1200 python -m pytest tests -m gpu -q > $OUT/c1_pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"
tail -15 $OUT/c1_pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/c1_bench_20.json 2> $OUT/c1_bench_20.err; echo "bench20 rc=$?"
timeout 300 python bench.py --no-extras > $OUT/c1_bench_default.json 2> $OUT/c1_bench_default.err; echo "bench default rc=$?"
cat $OUT/c1_bench_20.json | cut -c1-3000
cat $OUT/c1_bench_default.json | cut -c1-600
