#!/bin/bash
# D = 256 on 16x16x32 MFMAs (the pair kernel with PAIR = false, probe 540 / 541) vs the shipped 32x32x16 kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
FA_PP2=540,541 timeout 150 python cuda-learn-notes_amd/tools/fa_w4_probe.py 600 "1,2,256,256;2,3,512,256;4,8,2048,256;2,32,4096,256" > $OUT/fa_m16_d256.log 2>&1
grep -v amdgpu.ids $OUT/fa_m16_d256.log | grep "CHK\|^FA" | grep -v "w4 600"
