#!/bin/bash
# round 3, session bg: (1) k_extend8: slot -> priority mask through a 2-KB LDS table instead of three conditional swaps; (2) C2: refill threshold of the Cornell kernel under three pipelines
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:ab/base.so.bin lut:ab/lut.so.bin 2>&1 | tee $O/r03bg_ab_c5_perm_lut.log
cp ab/lut.so.bin single-file-vulkan-pathtracing_amd/libpt_amd.so
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bvh8" 2>&1 | tail -3 | tee $O/r03bg_pytest_bvh8_lut.txt
cp ab/base.so.bin single-file-vulkan-pathtracing_amd/libpt_amd.so
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" base:- r8:-:refill=8 r12:-:refill=12 r20:-:refill=20 r24:-:refill=24 r32:-:refill=32 2>&1 | tee $O/r03bg_ab_c2_refill.log
