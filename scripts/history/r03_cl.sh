#!/bin/bash
# round 3, session cl: `nt` cache policy on the streamed-once queue traffic (queue records, rays, hit records) -- parity subset, then C2 / C4 / C5 / C5x against plain loads and stores, interleaved
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cp ab/nt.so.bin single-file-vulkan-pathtracing_amd/libpt_amd.so
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "c1 or c4 or bvh8 or sample_groups or nee" 2>&1 | tail -3 | tee $O/r03cl_pytest.txt
AB_ROUNDS=6 AB_REPS=3 bash scripts/ab_env.sh "--steps 16 --warmup 2" plain:ab/base.so.bin nt:ab/nt.so.bin 2>&1 | tee $O/r03cl_ab_nt.log
AB_ROUNDS=3 AB_REPS=3 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" plain:ab/base.so.bin nt:ab/nt.so.bin 2>&1 | tee -a $O/r03cl_ab_nt.log
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" plain:ab/base.so.bin nt:ab/nt.so.bin 2>&1 | tee -a $O/r03cl_ab_nt.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" plain:ab/base.so.bin nt:ab/nt.so.bin 2>&1 | tee -a $O/r03cl_ab_nt.log
