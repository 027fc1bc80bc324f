#!/bin/bash
# round 3, session ae: two-level kernel, no EXIT marker entries: an instance visit ends at the stack height of its entry -- parity, A/B on C4
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "instance or c4 or inst or two_level or nee" > $O/r03ay_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03ay_pytest.txt
AB_ROUNDS=4 bash scripts/ab_env.sh "--config c4 --steps 8 --warmup 1" base:ab/base_c4.so.bin new:- 2>&1 | tee $O/r03ay_ab_c4_exit_restore.log
