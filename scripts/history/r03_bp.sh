#!/bin/bash
# round 3, session bp: tri_enter x refill once more on the final 8-wide kernel (no-spill instantiation, exact LDS stack)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:- e12r12:-:tri_enter=12 e20r12:-:tri_enter=20 e16r8:-:refill=8 e16r16:-:refill=16 e20r16:-:tri_enter=20,refill=16 e14r10:-:tri_enter=14,refill=10 e18r12:-:tri_enter=18 2>&1 | tee $O/r03bp_ab_c5_vote_again.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" base:- e20r12:-:tri_enter=20 e16r8:-:refill=8 e16r16:-:refill=16 2>&1 | tee -a $O/r03bp_ab_c5_vote_again.log
