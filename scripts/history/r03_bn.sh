#!/bin/bash
# round 3, session bn: instruction scheduler of the translation unit that holds k_extend8 (max-ILP since round 1, chosen for the BVH4 kernel): default / max-memory-clause / iterative-ilp on C5 and C5x
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" maxilp:ab/base.so.bin default:ab/sdef.so.bin memclause:ab/smem.so.bin iterilp:ab/siter.so.bin 2>&1 | tee $O/r03bn_ab_c5_scheduler.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" maxilp:ab/base.so.bin default:ab/sdef.so.bin 2>&1 | tee -a $O/r03bn_ab_c5_scheduler.log
