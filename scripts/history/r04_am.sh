#!/bin/bash
# round 4, GPU session AM: model of the 8-wide walk's pending-children policies on C5-like paths (CPU simulation over the device-built tree)
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python scripts/sim_bvh8_policies.py 1000000 600 2>&1 | tail -12 | tee $O/r04am_sim_bvh8_policies.log
