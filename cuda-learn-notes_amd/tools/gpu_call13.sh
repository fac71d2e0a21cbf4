#!/bin/bash
# key-split form of the 64-rows-per-wave attention kernel (probe 710/711) vs shipped at C4-like shapes
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
FA_PP2=700,710,711 timeout 400 python cuda-learn-notes_amd/tools/fa_w4_probe.py 608 "4,8,2048,64;1,48,8192,64;2,24,4096,64;8,8,1024,64;1,8,256,64;16,16,512,64;4,32,2048,64" > $OUT/fa_kvs.log 2>&1
grep -v amdgpu.ids $OUT/fa_kvs.log | grep "CHK\|^FA" 
