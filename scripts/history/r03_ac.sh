#!/bin/bash
# round 3, session ac: full GPU suite on the tree with the k_extend8 node-step change
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/r03ac_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03ac_pytest.txt
for s in "fuzz_trace.py 120 52000" "fuzz_instances.py 20 7700" "fuzz_render.py 150 9100"; do timeout 900 python scripts/$s 2>&1 | tail -2; done | tee $O/r03ac_fuzz.txt
