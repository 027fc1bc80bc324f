#!/bin/bash
# round 3, session ba: the vote of k_extend8 (tri_enter / tri_stay) on C5 and C5x; parity of the knobs first
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bvh8" 2>&1 | tail -3 | tee $O/r03ba_pytest_bvh8.txt
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:- e32:-:tri_enter=32,tri_stay=65 e24:-:tri_enter=24 e16:-:tri_enter=16 e8:-:tri_enter=8 e16s8:-:tri_enter=16,tri_stay=8 e16s4:-:tri_enter=16,tri_stay=4 e24s8:-:tri_enter=24,tri_stay=8 e64s8:-:tri_stay=8 e64s1:-:tri_stay=1 e12s4:-:tri_enter=12,tri_stay=4 2>&1 | tee $O/r03ba_ab_c5_vote.log
