#!/bin/bash
# round 3, session bm: the 8-wide kernel with the LDS stack sized to the tree, no-spill instantiation, 7 waves beyond the Infinity Cache -- parity, then C5 / C5x against the build before
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "bvh8 or ray_sorting or nee" 2>&1 | tail -3 | tee $O/r03bm_pytest.txt
AB_ROUNDS=2 AB_REPS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" before:ab/base.so.bin now:ab/fin.so.bin now7:ab/fin.so.bin:extend_blocks=7 2>&1 | tee $O/r03bm_ab_c5_c5x.log
AB_ROUNDS=2 AB_REPS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" before:ab/base.so.bin now:ab/fin.so.bin now6:ab/fin.so.bin:extend_blocks=6 2>&1 | tee -a $O/r03bm_ab_c5_c5x.log
