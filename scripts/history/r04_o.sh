#!/bin/bash
# round 4, GPU session O: k_shade with every queue record of a chunk requested up front (PT_SHADE_PRELOAD) on the configs whose
# shading gathers its tables from HBM (C5, C5x) and on C4 / C2 for the side effects.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
AB_ROUNDS=2 bash scripts/ab_many.sh "--config c5 --steps 4 --reps 3" build/shade_base.so.bin build/shade_preload.so.bin 2>&1 | tee $O/r04o_ab_shade_preload_c5.log
AB_ROUNDS=2 bash scripts/ab_many.sh "--config c5x --steps 2 --reps 3" build/shade_base.so.bin build/shade_preload.so.bin 2>&1 | tee $O/r04o_ab_shade_preload_c5x.log
AB_ROUNDS=2 bash scripts/ab_many.sh "--config c4 --steps 8 --reps 3" build/shade_base.so.bin build/shade_preload.so.bin 2>&1 | tee $O/r04o_ab_shade_preload_c4.log
AB_ROUNDS=2 bash scripts/ab_many.sh "--reps 5" build/shade_base.so.bin build/shade_preload.so.bin 2>&1 | tee $O/r04o_ab_shade_preload_c2.log
