#!/bin/bash
# round 3, session bc: smaller refill thresholds with the new vote of k_extend8 on C5 and C5x
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:- e16r16:-:tri_enter=16,refill=16 e16r12:-:tri_enter=16,refill=12 e16r8:-:tri_enter=16,refill=8 e16r4:-:tri_enter=16,refill=4 e12r12:-:tri_enter=12,refill=12 e12r8:-:tri_enter=12,refill=8 e20r8:-:tri_enter=20,refill=8 e16r1:-:tri_enter=16,refill=1 e16s8r8:-:tri_enter=16,tri_stay=8,refill=8 2>&1 | tee $O/r03bc_ab_c5_vote_refill.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" base:- e16r16:-:tri_enter=16,refill=16 e16r8:-:tri_enter=16,refill=8 e20r12:-:tri_enter=20,refill=12 e12r8:-:tri_enter=12,refill=8 2>&1 | tee $O/r03bc_ab_c5x_vote.log
