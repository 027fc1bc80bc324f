#!/bin/bash
# round 3, session ce: pipeline start rules (stagger 0 / 1 / 2) for C5, C5x and C4 on the final kernels
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:- st0:-:stagger=0 st1:-:stagger=1 p3st2:-:pipes=3,stagger=2 2>&1 | tee $O/r03ce_ab_stagger.log
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" base:- st0:-:stagger=0 st1:-:stagger=1 p3st2:-:pipes=3,stagger=2 2>&1 | tee -a $O/r03ce_ab_stagger.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" base:- st1:-:stagger=1 sb3:-:sort_bits=3 sb5:-:sort_bits=5 2>&1 | tee -a $O/r03ce_ab_stagger.log
