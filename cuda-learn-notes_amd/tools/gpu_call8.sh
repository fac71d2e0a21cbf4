#!/bin/bash
# after wiring hgemm_w4 into the dispatcher: hgemm GPU tests, C++ harness, bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_hgemm.py tests/test_scripts.py -m gpu -q -x > $OUT/c8_hgemm_tests.log 2>&1; echo "hgemm tests rc=$?"; tail -5 $OUT/c8_hgemm_tests.log
timeout 300 ./cuda-learn-notes_amd/harness/hgemm_bench 200 > $OUT/c8_hgemm_bench_cpp.log 2>&1; echo "harness rc=$?"
grep -v "max |err|" $OUT/c8_hgemm_bench_cpp.log | tail -40
timeout 600 python bench.py > $OUT/c8_bench.json 2> $OUT/c8_bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/c8_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","steps")}, d["roofline"], d["config"]["kernel"] if "kernel" in d["config"] else "")
print({k:v for k,v in d.items() if k.startswith("roofline_fa")})
PY
