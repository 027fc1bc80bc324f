#!/bin/bash
# round 4, GPU session J: the fused kernel's slot-batch policy (guided sizes on / off x batch scaled by sample groups 1 / 2 / 8):
# K = 16 / 2 / 1 through the fused leg and the world-8 shard, interleaved over two rounds.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=single-file-vulkan-pathtracing_amd/libpt_amd.so; cp $L /tmp/keep_j.so
for round in 1 2; do for v in g0s1 g1s1 g1s8 g0s8 g0s2; do
cp build/fused_$v.so.bin $L
python - $v <<'PY'
import importlib, json, sys, time, statistics
sys.path.insert(0, ".")
import bench
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
scene = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
film = pt.Film(ctx, 1920, 1080)
r = bench.fused_leg(pt, ctx, scene, film, 1920, 1080, 32, 8, 16, 0.0)
# rank 0 of world 8 at C3's size (32 frames)
f8 = pt.Film(ctx, 1920, 1080)
kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8, rank=0, world=8, pipeline=pt.PIPELINE_FUSED, frame=0, frame_count=32)
pt.render(scene, f8, pt.default_params(**kw))
ts = []
for _ in range(3):
    f8.clear(); t0 = time.perf_counter(); pt.render(scene, f8, pt.default_params(**kw)); ts.append((time.perf_counter() - t0) * 1e3)
print(sys.argv[1], "K16", r["k_steps"]["mrays_per_s"], "K2", r["k2"]["mrays_per_s"], "K1 ms", r["latency_1frame"]["median_ms"], "world-8 rank ms", round(statistics.median(ts), 2), flush=True)
PY
done; done 2>&1 | tee $O/r04j_fused_batch_policy.log
cp /tmp/keep_j.so $L
