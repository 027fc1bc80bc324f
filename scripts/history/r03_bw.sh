#!/bin/bash
# round 3, session bw: pipelines / shade rule / persistent blocks / refill of the Cornell kernel at the new default shape (16 frames x 8 groups)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" base:- p2:-:pipes=2 p4:-:pipes=4 st0:-:stagger=0 b6:-:extend_blocks=6 b5:-:extend_blocks=5 r8:-:refill=8 r24:-:refill=24 2>&1 | tee $O/r03bw_ab_c2_new_shape_knobs.log
