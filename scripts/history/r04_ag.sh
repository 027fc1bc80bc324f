#!/bin/bash
# round 4, GPU session AG: one-group fused launches hand their slots out tile-major (all frames of a tile, centre tiles first, border last):
# parity, then K = 16 / 8 / 4 on one device, the ranks of world 8 / 4 at the bench's 16 frames, C4
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "fused" 2>&1 | grep -E "passed|failed" | tee $O/r04ag_pytest.log
for K in 16 8 4; do
  AB_ROUNDS=2 bash scripts/ab_many.sh "--pipeline fused --steps $K --frames-in-flight $K --sample-groups 1 --reps 5" build/base.so.bin build/tilemajor.so.bin 2>&1 | cut -c1-60 | sed "s/^/K=$K groups=1 /" | tee -a $O/r04ag_ab_fused_tilemajor.log
done
AB_ROUNDS=2 bash scripts/ab_many.sh "--pipeline fused --config c4 --steps 8 --reps 3" build/base.so.bin build/tilemajor.so.bin 2>&1 | cut -c1-60 | sed "s/^/C4 /" | tee -a $O/r04ag_ab_fused_tilemajor.log
L=single-file-vulkan-pathtracing_amd/libpt_amd.so; cp $L /tmp/keep3.so
for B in base tilemajor; do
  cp build/$B.so.bin $L
  timeout 900 python scripts/probe_shard_efficiency.py 16 fused > $O/r04ag_shard_k16_fused_$B.json 2> $O/r04ag_shard_$B.err; sed "s/^/$B /" $O/r04ag_shard_$B.err | tee -a $O/r04ag_ab_fused_tilemajor.log
done
cp /tmp/keep3.so $L
