#!/bin/bash
# round 4, GPU session AX: C5, the 6-wave 8-wide kernel with / without its (never culling) bound, ALONE on the chip (one pipeline) and overlapped, and with shade priority 0
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c5 --steps 4" base_p1:build/base6.so.bin:pipes=1 nb_p1:build/nb6.so.bin:pipes=1 base_p2:build/base6.so.bin nb_p2:build/nb6.so.bin 2>&1 | cut -c1-60,150-200 | tee $O/r04ax_c5_nb_alone.log
