#!/bin/bash
# round 3, session aa: k_extend8 node step -- per-ray folded origin terms, branch-free hit mask -- parity, A/B on C5 / C5x
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "bvh8 or hbm8 or full_size or big_scene or c5_full or short_division or stadium or ploc" > $O/r03aa_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03aa_pytest.txt
AB_ROUNDS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:ab/base8.so.bin new:- 2>&1 | tee $O/r03aa_ab_c5_node8.log
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c5x --steps 2 --warmup 1" base:ab/base8.so.bin new:- 2>&1 | tee $O/r03aa_ab_c5x_node8.log
