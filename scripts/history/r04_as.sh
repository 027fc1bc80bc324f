#!/bin/bash
# round 4, GPU session AS: the fused kernel's shade-block threshold once more (after the tile-major hand-out), K = 16 and K = 1 shapes
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
AB_ROUNDS=2 AB_REPS=5 bash scripts/ab_env.sh "--pipeline fused" r32:-:refill=32 r36:-:refill=36 r40:-:refill=40 r44:-:refill=44 r48:-:refill=48 2>&1 | cut -c1-60 | tee $O/r04as_fused_refill.log
AB_ROUNDS=2 AB_REPS=5 bash scripts/ab_env.sh "--pipeline fused --steps 1" r32:-:refill=32 r40:-:refill=40 r48:-:refill=48 2>&1 | cut -c1-60 | sed 's/^/K=1 /' | tee -a $O/r04as_fused_refill.log
