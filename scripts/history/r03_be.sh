#!/bin/bash
# round 3, session be: k_extend8 with packed multiply-adds (near planes only / all / all at 5 waves) on C5 + parity; C4 joint sweep of refill / enter_min / leaf_min
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:ab/base.so.bin pknear:ab/pknear.so.bin pkfull:ab/pkfull.so.bin pkfull5:ab/pkfull5.so.bin 2>&1 | tee $O/r03be_ab_c5_pkfma.log
cp ab/pknear.so.bin single-file-vulkan-pathtracing_amd/libpt_amd.so
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bvh8" 2>&1 | tail -3 | tee $O/r03be_pytest_bvh8_pknear.txt
cp ab/pkfull5.so.bin single-file-vulkan-pathtracing_amd/libpt_amd.so
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bvh8" 2>&1 | tail -3 | tee $O/r03be_pytest_bvh8_pkfull5.txt
cp ab/base.so.bin single-file-vulkan-pathtracing_amd/libpt_amd.so
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" base:- r40:-:refill=40 r56:-:refill=56 r32:-:refill=32 r64:-:refill=64 l4:-:leaf_min=4 l12:-:leaf_min=12 l16:-:leaf_min=16 e8:-:enter_min=8 e12:-:enter_min=12 e24:-:enter_min=24 r32e8:-:refill=32,enter_min=8 r32l12:-:refill=32,leaf_min=12 r40e12l12:-:refill=40,enter_min=12,leaf_min=12 b5:-:inst16_blocks=5 b3:-:inst16_blocks=3 2>&1 | tee $O/r03be_ab_c4_joint.log
