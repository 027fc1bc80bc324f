#!/bin/bash
# round 3, session ch: waves per SIMD of the k_shade instantiation with its tables in HBM (64 VGPRs since the instancing template): 8 / 7 / 6 / 5 on C5 and C5x
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" w7:ab/base.so.bin w8:ab/h8.so.bin w6:ab/h6.so.bin w5:ab/h5.so.bin 2>&1 | tee $O/r03ch_ab_c5_shade_waves.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" w7:ab/base.so.bin w8:ab/h8.so.bin w5:ab/h5.so.bin 2>&1 | tee -a $O/r03ch_ab_c5_shade_waves.log
