#!/bin/bash
# round 3, session bk: k_extend8 without the spill path's address arithmetic when the tree's levels fit the LDS stack (ns), and the same at 7 waves per SIMD (72 VGPRs, 10 stack entries in LDS: ns7)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:ab/base.so.bin ns:ab/ns.so.bin ns_l10:ab/ns.so.bin:lds_stack=10 ns7_l10:ab/ns7.so.bin:lds_stack=10 ns7_l9:ab/ns7.so.bin:lds_stack=9 2>&1 | tee $O/r03bk_ab_c5_nospill_7waves.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" base:ab/base.so.bin ns:ab/ns.so.bin ns7_l10:ab/ns7.so.bin:lds_stack=10 2>&1 | tee -a $O/r03bk_ab_c5_nospill_7waves.log
cp ab/ns.so.bin single-file-vulkan-pathtracing_amd/libpt_amd.so
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bvh8" 2>&1 | tail -3 | tee $O/r03bk_pytest_bvh8.txt
PT_TUNE="lds_stack=2" timeout 600 python scripts/fuzz_trace.py 60 84000 2>&1 | tail -2 | tee -a $O/r03bk_pytest_bvh8.txt
