#!/bin/bash
# row sums on the matrix pipe (OPT_SUMM, probe 530 / 531) vs the shipped ping-pong attention kernel (500)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
FA_PP2=500,530,531 timeout 400 python cuda-learn-notes_amd/tools/fa_w4_probe.py 608 "4,8,2048,64;2,24,4096,64;8,8,1024,64;1,8,256,64" > $OUT/fa_summ.log 2>&1
FA_PP2=500,530 timeout 400 python cuda-learn-notes_amd/tools/fa_w4_probe.py 600 "4,8,2048,128;2,32,4096,128" >> $OUT/fa_summ.log 2>&1
grep -v amdgpu.ids $OUT/fa_summ.log | grep "CHK\|^FA" 
