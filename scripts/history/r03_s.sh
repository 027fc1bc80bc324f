#!/bin/bash
# round 3, session s: the live-count poll reads the count of eight rounds ago instead of draining both streams -- tests, A/B, timeline
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r03s_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03s_pytest.txt
AB_ROUNDS=4 bash scripts/ab_env.sh "--steps 16 --warmup 2" drain:ab/div_final.so.bin lagged:- 2>&1 | tee $O/r03s_ab_c2_lagged_poll.log
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" drain:ab/div_final.so.bin lagged:- 2>&1 | tee $O/r03s_ab_c4_lagged_poll.log
AB_ROUNDS=2 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" drain:ab/div_final.so.bin lagged:- 2>&1 | tee $O/r03s_ab_c5_lagged_poll.log
AB_ROUNDS=2 bash scripts/ab_env.sh "--steps 1 --warmup 2" drain:ab/div_final.so.bin lagged:- 2>&1 | tee $O/r03s_ab_c2_k1_lagged_poll.log
