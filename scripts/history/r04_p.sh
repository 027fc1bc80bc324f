#!/bin/bash
# round 4, GPU session P: k_shade asking for the three vertex rows of a 64-B record together with its fourth row
# (PT_SHADE_HOIST_REC) on the configs whose shading gathers the records from HBM (C5, C5x); parity of that build on the big-scene tests.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
AB_ROUNDS=3 bash scripts/ab_many.sh "--config c5 --steps 4 --reps 3" build/shade_base.so.bin build/shade_hoist.so.bin 2>&1 | tee $O/r04p_ab_shade_hoist_c5.log
AB_ROUNDS=3 bash scripts/ab_many.sh "--config c5x --steps 2 --reps 3" build/shade_base.so.bin build/shade_hoist.so.bin 2>&1 | tee $O/r04p_ab_shade_hoist_c5x.log
L=single-file-vulkan-pathtracing_amd/libpt_amd.so; cp $L /tmp/keep2.so; cp build/shade_hoist.so.bin $L
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hbm or big or soup or rec64 or large or 8wide or bvh8" 2>&1 | tail -5 | tee $O/r04p_hoist_parity.log
cp /tmp/keep2.so $L
