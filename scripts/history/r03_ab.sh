#!/bin/bash
# round 3, session ab: k_extend8 node-step variants (folded origin terms in all; carry hit mask c0/c1; v_mbcnt rank m0/m1) on C5
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
AB_ROUNDS=3 bash scripts/ab_env.sh "--config c5 --steps 4 --warmup 1" base:ab/base8.so.bin c0m0:ab/x8carry0.so.bin c1m0:ab/x8carry1.so.bin c0m1:ab/x8c0m1.so.bin c1m1:ab/x8c1m1.so.bin 2>&1 | tee $O/r03ab_ab_c5_node8_variants.log
