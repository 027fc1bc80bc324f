#!/bin/bash
# round 3, session cm: `nt` on the queue traffic of the instanced kernels only -- parity of the instanced tests, C4 and C2 against the build before
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cp ab/ntinst.so.bin single-file-vulkan-pathtracing_amd/libpt_amd.so
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_independent_pin.py -q -x -k "inst or c4 or two_level or nee or full_size or c1" 2>&1 | tail -3
 timeout 900 python scripts/fuzz_instances.py 12 15100 2>&1 | tail -2) | tee $O/r03cm_parity.txt
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" plain:ab/base.so.bin nt_inst:ab/ntinst.so.bin 2>&1 | tee $O/r03cm_ab_nt_inst.log
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" plain:ab/base.so.bin nt_inst:ab/ntinst.so.bin 2>&1 | tee -a $O/r03cm_ab_nt_inst.log
