#!/bin/bash
# round 4, GPU session AV: fused kernels without the per-slot initialisation of the spill heads (the slot's first pool entry ends its chain): parity (term-log tiers), K = 1 / 2
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "fused" 2>&1 | grep -E "passed|failed" | tee $O/r04av_pytest.log
AB_ROUNDS=3 AB_REPS=5 bash scripts/ab_env.sh "--pipeline fused --steps 1" base:build/nohead_base.so.bin new:build/nohead_new.so.bin 2>&1 | cut -c1-60 | sed 's/^/K=1 /' | tee $O/r04av_ab_nohead.log
AB_ROUNDS=2 AB_REPS=5 bash scripts/ab_env.sh "--pipeline fused --steps 2" base:build/nohead_base.so.bin new:build/nohead_new.so.bin 2>&1 | cut -c1-60 | sed 's/^/K=2 /' | tee -a $O/r04av_ab_nohead.log
