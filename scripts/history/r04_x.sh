#!/bin/bash
# round 4, GPU session X: both fused kernels with leaves of independent triangles: their tests; C2 fused unchanged
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "fused" 2>&1 | tail -15 | tee $O/r04x_pytest.log
timeout 600 python bench.py --pipeline fused --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c2 fused', d['value'], d['ms_per_step'])" | tee $O/r04x_c2_fused.log
