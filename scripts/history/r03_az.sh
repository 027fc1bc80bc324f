#!/bin/bash
# round 3, session az: k_shade walks runs of consecutive chunks (page locality of its ten streams): parity, A/B on C2 incl. the allocation lottery
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
L=single-file-vulkan-pathtracing_amd/libpt_amd.so; cp $L /tmp/keep.so
for v in run1 run4 run16 run64 run1 run16; do cp ab/$v.so.bin $L; echo -n "$v: "; python scripts/probe_alloc_modes.py 6 | cut -c60-260; done 2>&1 | tee $O/r03az_shade_runs.log
cp ab/run16.so.bin $L; timeout 900 python -m pytest tests -m gpu -x -q -k "c1 or c2 or full_size or group or render" 2>&1 | tail -2
cp /tmp/keep.so $L
