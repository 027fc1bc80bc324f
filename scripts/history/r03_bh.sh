#!/bin/bash
# round 3, session bh: k_extend8 without the (unused) repeat loop around the triangle step -- A/B on C5 / C5x; then the full GPU suite on it
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" loop:ab/base.so.bin noloop:ab/n2.so.bin 2>&1 | tee $O/r03bh_ab_c5_tri_loop.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" loop:ab/base.so.bin noloop:ab/n2.so.bin 2>&1 | tee -a $O/r03bh_ab_c5_tri_loop.log
cp ab/n2.so.bin single-file-vulkan-pathtracing_amd/libpt_amd.so
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/r03bh_pytest_gpu_suite.txt
cat $O/r03bh_pytest_gpu_suite.txt
