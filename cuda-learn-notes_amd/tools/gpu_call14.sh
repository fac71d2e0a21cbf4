#!/bin/bash
# life stamps of every wave of the shipped D = 64 attention kernel at config C4
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 200 python cuda-learn-notes_amd/tools/fa_life_stamps.py 4,8,2048,64 > $OUT/fa_life.log 2>&1
timeout 200 python cuda-learn-notes_amd/tools/fa_life_stamps.py 4,8,4096,64 >> $OUT/fa_life.log 2>&1
grep -v amdgpu.ids $OUT/fa_life.log | tail -100
