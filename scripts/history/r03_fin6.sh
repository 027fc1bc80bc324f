#!/bin/bash
# round 3, last session: the full GPU suite, then the measurement pass of scripts/r03_final.sh on the final code
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/r03fin6_pytest_gpu_suite.txt
cat $O/r03fin6_pytest_gpu_suite.txt
bash scripts/r03_final.sh r03fin6
