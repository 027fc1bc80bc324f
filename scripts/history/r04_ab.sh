#!/bin/bash
# round 4, GPU session AB: the whole GPU suite on the tree with the two-launch surface-area builder
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04ab_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $O/r04ab_pytest.log | tail -3
