#!/bin/bash
# round 3, session z: k_shade variants (queue-record preload, items per thread) under the free-running three-pipeline schedule
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" cur:- preload7:ab/preload7.so.bin preload5:ab/preload5.so.bin items4w5:ab/items4w5.so.bin 2>&1 | tee $O/r03z_ab_c2_shade_variants.log
