#!/bin/bash
# round 4, GPU session AT: C5 / C5x with ONE pipeline: the traversal and the shading kernel alone on the chip (what a fused kernel for HBM scenes could reach)
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
AB_ROUNDS=1 bash scripts/ab_env.sh "--config c5 --steps 4" p2:- p1:-:pipes=1 p3:-:pipes=3 2>&1 | cut -c1-200 | tee $O/r04at_c5_pipes.log
AB_ROUNDS=1 bash scripts/ab_env.sh "--config c5x --steps 2" p2:- p1:-:pipes=1 2>&1 | cut -c1-200 | sed 's/^/c5x /' | tee -a $O/r04at_c5_pipes.log
