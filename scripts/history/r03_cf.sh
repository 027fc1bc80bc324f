#!/bin/bash
# round 3, session cf: k_shade with instancing as a template parameter (the single-level instantiation: 70 VGPRs, no spills) and the
# per-(instance, triangle) normal + tangent table -- parity on instanced scenes, then C4 and C2 against the build before
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_independent_pin.py -q -x -k "inst or c4 or two_level or nee or full_size" 2>&1 | tail -3
 timeout 900 python scripts/fuzz_instances.py 20 13900 2>&1 | tail -2
 PT_TUNE="inst_frames=0" timeout 900 python scripts/fuzz_instances.py 10 14000 2>&1 | tail -2) | tee $O/r03cf_parity.txt
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" base:ab/base.so.bin if6:ab/if6.so.bin if6_off:ab/if6.so.bin:inst_frames=0 if7:ab/if7.so.bin if5:ab/if5.so.bin 2>&1 | tee $O/r03cf_ab_c4_inst_frames.log
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" base:ab/base.so.bin new:ab/if6.so.bin 2>&1 | tee $O/r03cf_ab_c2_shade_template.log
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:ab/base.so.bin new:ab/if6.so.bin 2>&1 | tee -a $O/r03cf_ab_c2_shade_template.log
