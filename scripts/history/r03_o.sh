#!/bin/bash
# round 3, session o: final form of the exact short division (+ primitive ids on ties, v_mbcnt rank): full GPU suite, A/B vs the build before
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r03o_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03o_pytest.txt
AB_ROUNDS=4 bash scripts/ab_env.sh "--steps 16 --warmup 2" base:ab/base_before_div.so.bin new:- 2>&1 | tee $O/r03o_ab_c2_exact_div.log
AB_ROUNDS=3 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" base:ab/base_before_div.so.bin new:- 2>&1 | tee $O/r03o_ab_c4_exact_div.log
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:ab/base_before_div.so.bin new:- 2>&1 | tee $O/r03o_ab_c5_exact_div.log
