#!/bin/bash
# round 4, GPU session R: k_fused_inst -- parity against the wavefront pipeline; then config C4 over refill, the TLAS share of LDS,
# the LDS stack depth and the block shape (512 x 2 per CU, 1024 x 1, 640 x 2 at 96 VGPRs).
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python scripts/probe_fused_inst.py 2>&1 | tail -20 | tee $O/r04r_fused_inst_parity.log
A="--config c4 --pipeline fused --steps 8"
AB_ROUNDS=1 bash scripts/ab_env.sh "$A" base:- r32:-:refill=32 r40:-:refill=40 r56:-:refill=56 r64:-:refill=64 \
   t0:-:tlas_lds_kb=0 t16:-:tlas_lds_kb=16 t20:-:tlas_lds_kb=20 s12t28:-:lds_stack=12,tlas_lds_kb=28 s10t36:-:lds_stack=10,tlas_lds_kb=36 \
   k1024t8:build/fi1024.so.bin k1024t32:build/fi1024.so.bin:tlas_lds_kb=32 k1024t48:build/fi1024.so.bin:tlas_lds_kb=48 \
   k1024s12t64:build/fi1024.so.bin:lds_stack=12,tlas_lds_kb=64 k640w5:build/fi640w5.so.bin:lds_stack=12 \
   e8:-:enter_min=8 e32:-:enter_min=32 l16:-:leaf_min=16 y0:-:node_yield=0 2>&1 | tee $O/r04r_fused_inst_sweep.log
