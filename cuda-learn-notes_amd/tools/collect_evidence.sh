#!/bin/bash
# Copy the judged subset of a round's gpurun_out/ files into profiles/ (tracked) and take the CPU-suite log of THIS tree beside the GPU log
# (round 5's last kernel commit landed after its last evidence call and broke the CPU suite unnoticed).
#   collect_evidence.sh <tag, e.g. r06> [--no-cpu]
TAG=${1:-r06}
REPO=$(cd "$(dirname "$0")/../.." && pwd); OUT=$REPO/gpurun_out; P=$REPO/profiles
for f in pytest_gpu.log bench_20steps.json bench_20steps.log bench_default.json bench_default.log bench_detail_20steps.json bench_detail_default.json bench_kernel_stats.csv \
         pmc_hgemm.json pmc_fa_d64.json pmc_fa_d128.json pmc_fa_d256.json pmc_fa_d512.json pmc_fa_d768_dw4.json pmc_fa_d768_dring.json pmc_fa_d1024_dw4.json pmc_fa_d1024_dring.json \
         fa_kernel_trace.csv bw_rocprof.json bw_rocprof.txt reference_style_scripts_on_gpu.log hgemm_bench_cpp.log fa_dw4_probe_final.log hgemm_splitk_fused_probe_evidence.log \
         host_call_overhead_final.log scripts_vs_torch.log profile_round.log fa_c4_stamps.log fa_tol_calibration.log stream_forms_ubench.log transpose_forms_ubench.log \
         fa_stage1_vs_stage2.log hipblaslt_probe.log determinism_stress.log fa_ck_tile_comparator.log fa_one_stage_probe.log hgemm_w4s_probe.log hgemm_reference_sweep.log \
         hgemm_tail_probe_after.log hgemm_rect_probe.log fa_small_grid_probe.log fa_fscale_probe.log fa_back_to_back_stress.log smoke.log rung_survey.log sgemm_reference_sweep.log; do
  [ -s $OUT/${TAG}_$f ] && cp $OUT/${TAG}_$f $P/${TAG}_$f
done
if [ "$2" != "--no-cpu" ]; then
  ( cd $REPO && git rev-parse HEAD && git status --short | head -20 && timeout 2400 python -m pytest tests -q -m "not gpu" 2>&1 | tail -6 ) > $P/${TAG}_pytest_cpu.log 2>&1
  tail -2 $P/${TAG}_pytest_cpu.log
fi
ls $P | grep -c "^${TAG}_"
