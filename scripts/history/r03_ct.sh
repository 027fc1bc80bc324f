#!/bin/bash
# round 3, session ct: nt on the 8-wide kernel's ray loads alone (the rays are read once, through a sorted permutation on C5x) -- C5 and C5x
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" plain:ab/base.so.bin nt_ld:ab/e8ld.so.bin 2>&1 | tee $O/r03ct_ab_e8_nt_loads.log
AB_ROUNDS=3 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" plain:ab/base.so.bin nt_ld:ab/e8ld.so.bin 2>&1 | tee -a $O/r03ct_ab_e8_nt_loads.log
