#!/bin/bash
# 16x16x32 attention kernel: 64-rows-per-wave form (545 / 546) vs the shipped 512-row kernel (700); D = 128 form (540 / 542) vs shipped (500)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
FA_PP2=540,545,546,700 timeout 400 python cuda-learn-notes_amd/tools/fa_w4_probe.py 608 "1,8,512,64;1,48,8192,64;2,32,4096,64;1,16,16384,64" > $OUT/fa_m16b.log 2>&1
FA_PP2=500,540,542 timeout 400 python cuda-learn-notes_amd/tools/fa_w4_probe.py 600 "1,8,256,128;4,8,2048,128;2,32,4096,128;1,24,8192,128" >> $OUT/fa_m16b.log 2>&1
grep -v amdgpu.ids $OUT/fa_m16b.log | grep "CHK\|^FA\|Error\|error" | grep -v "sdpa\|w4 60"
