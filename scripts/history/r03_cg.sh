#!/bin/bash
# round 3, session cg: instanced k_shade at 5 / 4 waves, with and without the table; then the full GPU suite on the tree's build (5 waves)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" base:ab/base.so.bin if5:ab/if5.so.bin if5_off:ab/if5.so.bin:inst_frames=0 if4:ab/if4.so.bin if5_b5:ab/if5.so.bin:inst16_blocks=5 if5_b3:ab/if5.so.bin:inst16_blocks=3 2>&1 | tee $O/r03cg_ab_c4_shade_waves.log
cp ab/if5.so.bin single-file-vulkan-pathtracing_amd/libpt_amd.so
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | tee $O/r03cg_pytest_gpu_suite.txt
