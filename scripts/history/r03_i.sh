#!/bin/bash
# round 3, session i: full GPU suite (after the C5 test learnt that AUTO = 8-wide tree beyond L2)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/r03i_pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/r03i_pytest.txt; grep -n "Error" -B2 -A6 $O/r03i_pytest.txt | head -60
