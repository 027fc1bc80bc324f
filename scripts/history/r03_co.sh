#!/bin/bash
# round 3, session co: `nt` on k_shade's queue stores for every scene class (A; instanced kernels keep nt on all their queue traffic) -- C2 / C4 / C5 / C5x against the build before; B = shade stores only, everywhere
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=6 AB_REPS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" before:ab/base.so.bin A:ab/A.so.bin 2>&1 | tee $O/r03co_ab_nt_shade_stores.log
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" before:ab/base.so.bin A:ab/A.so.bin B:ab/B.so.bin 2>&1 | tee -a $O/r03co_ab_nt_shade_stores.log
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" before:ab/base.so.bin A:ab/A.so.bin 2>&1 | tee -a $O/r03co_ab_nt_shade_stores.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" before:ab/base.so.bin A:ab/A.so.bin 2>&1 | tee -a $O/r03co_ab_nt_shade_stores.log
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--steps 2 --warmup 1" before:ab/base.so.bin A:ab/A.so.bin 2>&1 | tee -a $O/r03co_ab_nt_shade_stores.log
