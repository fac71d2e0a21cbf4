#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the bench command + separate PMC passes for the
# headline kernels (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass; PMC never with
# sys/hip/hsa trace domains). Usage: profile_round.sh <round-tag, e.g. r01>
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
T=$REPO/cuda-learn-notes_amd/tools
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o $TAG -- python $REPO/bench.py --steps 300 --warmup 100 --no-extras > $OUT/prof_bench.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/prof_bench/${TAG}_kernel_stats.csv")))
with open("$OUT/${TAG}_bench_kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(["Name(120)", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows: w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
pmc() {  # name, counters..., then "--", then target args
  local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --output-format csv --pmc "${ctrs[@]}" -d $OUT/pmc_$name -o pmc -- python $T/prof_target.py "$@" > $OUT/pmc_$name.log 2>&1
}
HG="hgemm 14 0 1 64 203 4096 12"  # kind 14, variant 203 = production schedule 26 with the production epilogue 3 (non-temporal C stores; csrc/hgemm.hip W4_PRODUCTION, W4_EPILOGUE)
pmc hg_fetch FETCH_SIZE -- $HG
pmc hg_write WRITE_SIZE -- $HG
pmc hg_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- $HG
python $T/pmc_summary.py hgemm $OUT/${TAG}_pmc_hgemm.json $OUT/pmc_hg_fetch $OUT/pmc_hg_write $OUT/pmc_hg_sq > /dev/null
for D in 64 128 256; do  # 256: the production fa2_fwd_m16_pair_kernel (no trace / PMC pass of it existed in round 2)
  FA="fa 4 8 2048 $D 2 12"
  pmc fa${D}_fetch FETCH_SIZE -- $FA
  pmc fa${D}_write WRITE_SIZE -- $FA
  pmc fa${D}_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- $FA
  python $T/pmc_summary.py fa2 $OUT/${TAG}_pmc_fa_d$D.json $OUT/pmc_fa${D}_fetch $OUT/pmc_fa${D}_write $OUT/pmc_fa${D}_sq > /dev/null
done
# config C5 (D = 512)
FA="fa 1 32 4096 512 2 6"
pmc fa512_fetch FETCH_SIZE -- $FA
pmc fa512_write WRITE_SIZE -- $FA
pmc fa512_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- $FA
python $T/pmc_summary.py fa2 $OUT/${TAG}_pmc_fa_d512.json $OUT/pmc_fa512_fetch $OUT/pmc_fa512_write $OUT/pmc_fa512_sq > /dev/null
ls -la $OUT/${TAG}_*
