#!/bin/bash
# round 4, GPU session AP: k_extend8 without the pending-group bound x 6 / 7 waves per SIMD on C5 and C5x
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
AB_ROUNDS=3 bash scripts/ab_env.sh "--config c5 --steps 4" base6:build/e8_base.so.bin:extend_blocks=6 nb6:build/e8_nb.so.bin:extend_blocks=6 base7:build/e8_base.so.bin:extend_blocks=7 nb7:build/e8_nb.so.bin:extend_blocks=7 2>&1 | cut -c1-150 | tee $O/r04ap_ab_e8_nb_waves_c5.log
AB_ROUNDS=3 bash scripts/ab_env.sh "--config c5x --steps 2" base6:build/e8_base.so.bin:extend_blocks=6 nb6:build/e8_nb.so.bin:extend_blocks=6 base7:build/e8_base.so.bin:extend_blocks=7 nb7:build/e8_nb.so.bin:extend_blocks=7 2>&1 | cut -c1-150 | tee $O/r04ap_ab_e8_nb_waves_c5x.log
