#!/bin/bash
# round 4, GPU session K: fused kernel after the batch-policy decision -- tests, the fused leg, and the sample-group count at K = 1, 2, 4.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "fused or c3_full" > $O/r04k_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r04k_pytest.log
python - <<'PY' 2>&1 | tee gpurun_out/r04k_fused_groups.log
import importlib, json, sys, time, statistics
sys.path.insert(0, ".")
import bench
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
scene = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
film = pt.Film(ctx, 1920, 1080)
print("fused leg:", json.dumps(bench.fused_leg(pt, ctx, scene, film, 1920, 1080, 32, 8, 16, 0.0)), flush=True)
for K in (1, 2, 4, 8):
    for g in (1, 2, 4, 8, 16, 32):
        f2 = pt.Film(ctx, 1920, 1080)
        kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, frame=0, frame_count=K, sample_groups=g)
        pt.render(scene, f2, pt.default_params(**kw))
        ts = []
        for _ in range(4):
            f2.clear(); t0 = time.perf_counter(); pt.render(scene, f2, pt.default_params(**kw)); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"K {K} groups {g:2d}: {statistics.median(ts):7.3f} ms total, {statistics.median(ts) / K:6.3f} ms per frame, workspace {ctx.stats().workspace_bytes / 2**30:.2f} GB", flush=True)
        f2.close()
PY
