#!/bin/bash
# ping-pong attention kernel on 16x16x32 MFMAs (probe 540 / 541 / 542) vs the shipped 32x32x16 kernel (500)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
FA_PP2=500,540,541,542 timeout 400 python cuda-learn-notes_amd/tools/fa_w4_probe.py 608 "1,8,256,64;4,8,2048,64;2,24,4096,64;8,8,1024,64;1,48,8192,64" > $OUT/fa_m16.log 2>&1
grep -v amdgpu.ids $OUT/fa_m16.log | grep "CHK\|^FA\|Error\|error" 
