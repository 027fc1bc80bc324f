#!/bin/bash
# round 3, session bo: two-level kernel, pushes and pop loop without the spill test while the whole wave is within the LDS stack -- parity on instanced scenes (small LDS stacks force the slow path too), then C4 A/B
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cp ab/roomy.so.bin single-file-vulkan-pathtracing_amd/libpt_amd.so
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "inst or c4 or two_level or nee" 2>&1 | tail -3
 timeout 900 python scripts/fuzz_instances.py 20 11000 2>&1 | tail -2
 PT_TUNE="lds_stack=3" timeout 900 python scripts/fuzz_instances.py 12 11500 2>&1 | tail -2
 PT_TUNE="lds_stack=6" timeout 900 python scripts/fuzz_instances.py 12 11600 2>&1 | tail -2) | tee $O/r03bo_parity.txt
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" base:ab/base.so.bin roomy:ab/roomy.so.bin 2>&1 | tee $O/r03bo_ab_c4_roomy.log
