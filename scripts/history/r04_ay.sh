#!/bin/bash
# round 4, GPU session AY: C5 overlapped: can the 6-wave kernel without its bound (+3.3 % alone) be made to pay in the two-pipeline frame? refill / grid sweeps
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c5 --steps 4" base:build/base6.so.bin nb:build/nb6.so.bin nb_r8:build/nb6.so.bin:refill=8 nb_r16:build/nb6.so.bin:refill=16 nb_r24:build/nb6.so.bin:refill=24 base_r16:build/base6.so.bin:refill=16 nb_te8:build/nb6.so.bin:tri_enter=8 nb_te24:build/nb6.so.bin:tri_enter=24 nb_ls12:build/nb6.so.bin:lds_stack=12 2>&1 | cut -c1-60 | tee $O/r04ay_c5_nb_sweep.log
