#!/bin/bash
# round 3, session cj: waves per SIMD of the single-level k_shade (70 VGPRs, no spills since the instancing template): 7 / 6 / 5, and 5 with all queue records requested up front
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" w7:ab/base.so.bin w6:ab/sw6.so.bin w5:ab/sw5.so.bin w5pre:ab/sw6p.so.bin 2>&1 | tee $O/r03cj_ab_c2_shade_waves.log
