#!/bin/bash
# round 4, GPU session AH: where a small fused launch's time goes (wall / device / kernel), K = 16 and 32, rank 0 of world 1 .. 8
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python scripts/probe_fused_tail.py 16 32 4 2>&1 | tee $O/r04ah_fused_tail.log
