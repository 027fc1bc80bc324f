#!/bin/bash
# round 3, session ck: single-level k_shade at 7 against 6 waves per SIMD, ten interleaved processes each (the allocation levels are the noise to beat)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=10 AB_REPS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" w7:ab/base.so.bin w6:ab/sw6.so.bin 2>&1 | tee $O/r03ck_ab_c2_shade_7_vs_6_waves.log
