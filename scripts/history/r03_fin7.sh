#!/bin/bash
# round 3, last session, final code: the full GPU suite, then the measurement pass of scripts/r03_final.sh
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/r03fin7_pytest_gpu_suite.txt
cat $O/r03fin7_pytest_gpu_suite.txt
bash scripts/r03_final.sh r03fin7
