#!/bin/bash
# round 3, session bb: refill x tri_enter of k_extend8 on C5, then the best few on C5x
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:- e16:-:tri_enter=16 e16r24:-:tri_enter=16,refill=24 e16r16:-:tri_enter=16,refill=16 e16r40:-:tri_enter=16,refill=40 e12r24:-:tri_enter=12,refill=24 e12r16:-:tri_enter=12,refill=16 e12s4r24:-:tri_enter=12,tri_stay=4,refill=24 e20:-:tri_enter=20 e14:-:tri_enter=14 e16l8:-:tri_enter=16,lds_stack=8 e16l16:-:tri_enter=16,lds_stack=16 2>&1 | tee $O/r03bb_ab_c5_vote_refill.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" base:- e16:-:tri_enter=16 e12s4:-:tri_enter=12,tri_stay=4 e16r24:-:tri_enter=16,refill=24 e24:-:tri_enter=24 2>&1 | tee $O/r03bb_ab_c5x_vote.log
