set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out; T=$REPO/cuda-learn-notes_amd/tools
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA"
for cfg in "64 8 13 0" "64 8 13 100" "64 8 13 1" "64 8 13 7" "128 8 15 0"; do
  tag=$(echo $cfg | tr ' ' '_')
  HH=48; NN=8192; if [ "${cfg%% *}" = "128" ]; then HH=32; NN=4096; fi
  rocprofv3 --kernel-trace --output-format csv --pmc $P1 -d $OUT/pmcfa_${tag}_p1 -o pmc -- python $T/prof_target.py fa2 $cfg 1 $HH $NN 6 > $OUT/pmcfa_${tag}_p1.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc $P2 -d $OUT/pmcfa_${tag}_p2 -o pmc -- python $T/prof_target.py fa2 $cfg 1 $HH $NN 6 > $OUT/pmcfa_${tag}_p2.log 2>&1
  python $T/pmc_summary.py fa2_fwd $OUT/pmcfa_${tag}.json $OUT/pmcfa_${tag}_p1 $OUT/pmcfa_${tag}_p2 > /dev/null
done
