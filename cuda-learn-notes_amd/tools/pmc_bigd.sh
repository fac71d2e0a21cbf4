# PMC passes for the big-D attention kernel variants at config C5.   bash pmc_bigd.sh "201 204"
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out; T=$REPO/cuda-learn-notes_amd/tools
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA"
for abl in ${1:-201 204}; do
  tag=bigd_${abl}
  rocprofv3 --kernel-trace --output-format csv --pmc $P1 -d $OUT/pmc_${tag}_p1 -o pmc -- python $T/prof_target.py fa2 512 4 15 $abl 1 32 4096 4 > $OUT/pmc_${tag}_p1.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc $P2 -d $OUT/pmc_${tag}_p2 -o pmc -- python $T/prof_target.py fa2 512 4 15 $abl 1 32 4096 4 > $OUT/pmc_${tag}_p2.log 2>&1
  python $T/pmc_summary.py fa2_fwd $OUT/pmc_${tag}.json $OUT/pmc_${tag}_p1 $OUT/pmc_${tag}_p2
done
