#!/bin/bash
# round 3, session bf: (1) the BVH4 HBM kernel's vote / refill on C5 (--extend hbm); (2) C5 8-wide: pipelines, ray sorting, batch shapes with the new vote
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --extend hbm --steps 4 --warmup 1" base:- e16:-:tri_enter=16 e16r16:-:tri_enter=16,refill=16 e16r12:-:tri_enter=16,refill=12 e24r16:-:tri_enter=24,refill=16 e12r12:-:tri_enter=12,refill=12 e8r8:-:tri_enter=8,refill=8 r16:-:refill=16 2>&1 | tee $O/r03bf_ab_c5_bvh4_vote.log
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:- p3:-:pipes=3 p1:-:pipes=1 2>&1 | tee $O/r03bf_ab_c5_pipes.log
for a in "--sort-rays on" "--sample-groups 8" "--sample-groups 32" "--frames-in-flight 2 --sample-groups 16" "--frames-in-flight 4 --sample-groups 8"; do
  echo "== $a"; AB_ROUNDS=1 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1 $a" x:-
done 2>&1 | tee $O/r03bf_c5_shapes.log
