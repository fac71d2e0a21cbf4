#!/bin/bash
# D = 512 pair kernel on 16x16x32 as the C5 production path: probe, the D = 512 tests, PMC passes of the new kernel,
# the two bench lines, the C5 trace row. Stops at the first failure.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
OUT=$REPO/gpurun_out; T=$REPO/cuda-learn-notes_amd/tools; mkdir -p $OUT
FA_PP2=220,540 timeout 100 python $T/fa_w4_probe.py 210 "2,3,256,512;1,32,4096,512" > $OUT/fa_m16_pair2.log 2>&1
grep -v amdgpu.ids $OUT/fa_m16_pair2.log | grep "CHK\|^FA" | grep -v "sdpa\|ERR"
timeout 200 python -m pytest tests/test_gpu_flash_attn.py -m gpu -x -q --timeout 90 -k "512 or tiling or golden or c5 or large" > $OUT/c25_tests.log 2>&1 || { echo "tests failed"; tail -15 $OUT/c25_tests.log; exit 8; }
tail -2 $OUT/c25_tests.log
( cd /tmp && export TMPDIR=/tmp
  pmc() { local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
    timeout 90 rocprofv3 --kernel-trace --output-format csv --pmc "${ctrs[@]}" -d $OUT/pmc_$name -o pmc -- python $T/prof_target.py "$@" > $OUT/pmc_$name.log 2>&1; }
  FA="fa 1 32 4096 512 2 6"
  rm -rf $OUT/pmc_fa512_fetch $OUT/pmc_fa512_write $OUT/pmc_fa512_sq
  pmc fa512_fetch FETCH_SIZE -- $FA
  pmc fa512_write WRITE_SIZE -- $FA
  pmc fa512_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- $FA
  python $T/pmc_summary.py fa2 $OUT/r02_pmc_fa_d512.json $OUT/pmc_fa512_fetch $OUT/pmc_fa512_write $OUT/pmc_fa512_sq > /dev/null && cp $OUT/r02_pmc_fa_d512.json $REPO/profiles/ )
head -c 300 $OUT/r02_pmc_fa_d512.json; echo
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r02_bench_20steps.json 2> $OUT/r02_bench_20steps.err || { echo bench20 failed; exit 7; }
timeout 150 python bench.py > $OUT/r02_bench_default.json 2> $OUT/r02_bench_default.err || { echo bench failed; exit 7; }
python - <<PY
import json
for f in ("r02_bench_20steps.json","r02_bench_default.json"):
    d=json.loads(open("$OUT/"+f).read().strip().splitlines()[-1]); c=d["config"]
    r=d["roofline_fa2_c5_d512"]
    print(f, d["value"], d["roofline"]["frac"], c["settle_ms_per_step"], "| C4", d["roofline_fa2_c4_d64"]["achieved"], "d128", d["roofline_fa2_d128"]["achieved"], "c5", r["achieved"], r["traffic"], r["mfma_busy"], r["kernel"][:30])
PY
( cd /tmp && export TMPDIR=/tmp; timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fatrace_c5 -o t -- python $T/prof_target.py fa 1 32 4096 512 2 20 > $OUT/fatrace_c5.log 2>&1; grep "fa2" $OUT/fatrace_c5/t_kernel_stats.csv | cut -c1-200 )
