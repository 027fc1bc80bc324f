#!/bin/bash
# round 4, GPU session G: 8-wide node decode through v_perm / v_fma_mix (A/B on C5 and C5x + parity of the variant), the wavefront
# C2 shape against its workspace (2 / 4 / 8 sample groups, three processes each), fused shards under the new shape rule.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=single-file-vulkan-pathtracing_amd/libpt_amd.so; cp $L /tmp/keep_g.so
cp build/node8_perm.so.bin $L
timeout 600 python -m pytest tests -m gpu -x -q -k "bvh8 or soup or hbm8 or full_size_frames" > $O/r04g_pytest_perm.log 2>&1; echo "pytest(perm) rc=$?"; tail -2 $O/r04g_pytest_perm.log
cp /tmp/keep_g.so $L
AB_ROUNDS=3 bash scripts/ab_many.sh "--config c5 --steps 4 --reps 3" build/node8_base.so.bin build/node8_perm.so.bin 2>&1 | tee $O/r04g_ab_node8_c5.log
AB_ROUNDS=3 bash scripts/ab_many.sh "--config c5x --steps 2 --reps 3" build/node8_base.so.bin build/node8_perm.so.bin 2>&1 | tee $O/r04g_ab_node8_c5x.log
cp /tmp/keep_g.so $L
for round in 1 2 3; do for g in 2 4 8; do
  python bench.py --sample-groups $g --frames-in-flight 16 --no-extra-legs --no-cpu-baseline --reps 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('groups $g', d['value'], d['value_min'], d['value_max'], 'ws GB', round(d['workspace_bytes']/2**30,1))"
done; done | tee $O/r04g_c2_shapes.log
timeout 900 python scripts/probe_shard_efficiency.py 32 fused > $O/r04_shard_efficiency_fused.json 2> $O/r04_shard_efficiency_fused.err; cat $O/r04_shard_efficiency_fused.err
