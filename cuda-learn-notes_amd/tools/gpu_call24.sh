#!/bin/bash
# D = 512 pair kernel on 16x16x32 MFMAs (probe 540 / 541) vs the shipped d-split kernel (210 / 220) and SDPA
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
FA_PP2=220,540,541 timeout 200 python cuda-learn-notes_amd/tools/fa_w4_probe.py 210 "1,2,128,512;2,3,256,512;1,32,4096,512" > $OUT/fa_m16_pair.log 2>&1
grep -v amdgpu.ids $OUT/fa_m16_pair.log | grep "CHK\|^FA\|Error\|error"
