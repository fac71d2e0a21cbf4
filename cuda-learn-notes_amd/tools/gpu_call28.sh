#!/bin/bash
# the 16x16x32 pair kernel with the scores scaled in fp32 (probe 544) vs its fp16-pre-scaled form (540) and the shipped kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
FA_PP2=540,544 timeout 100 python cuda-learn-notes_amd/tools/fa_w4_probe.py 600 "2,3,512,256;4,8,2048,256;2,32,4096,256" > $OUT/fa_m16_f32scale.log 2>&1
FA_PP2=220,540,544 timeout 100 python cuda-learn-notes_amd/tools/fa_w4_probe.py 210 "2,3,256,512;1,32,4096,512" >> $OUT/fa_m16_f32scale.log 2>&1
grep -v amdgpu.ids $OUT/fa_m16_f32scale.log | grep "CHK\|^FA" | grep -v "w4 600\|sdpa\|ERR"
