#!/bin/bash
# round 4, GPU session L: one slot counter against eight (one per XCD's share of the workgroups, work stealing): fused tests on the
# new kernel, then both builds interleaved -- K = 16 / 2 / 1 by sample groups, and rank 0 of world 8 at C3's size.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "fused or c3_full" > $O/r04l_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r04l_pytest.log
L=single-file-vulkan-pathtracing_amd/libpt_amd.so; cp $L /tmp/keep_l.so
for round in 1 2; do for v in one_counter xcd_counters; do
cp build/fused_$v.so.bin $L
python - $v <<'PY'
import importlib, sys, time, statistics
sys.path.insert(0, ".")
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
scene = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
out = []
for K, g, world in ((16, 1, 1), (8, 1, 1), (4, 8, 1), (4, 16, 1), (2, 8, 1), (2, 16, 1), (2, 32, 1), (1, 16, 1), (1, 32, 1), (32, 1, 8)):
    f2 = pt.Film(ctx, 1920, 1080)
    kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, frame=0, frame_count=K, sample_groups=g, rank=0, world=world)
    pt.render(scene, f2, pt.default_params(**kw))
    ts = []
    for _ in range(5):
        f2.clear(); t0 = time.perf_counter(); pt.render(scene, f2, pt.default_params(**kw)); ts.append((time.perf_counter() - t0) * 1e3)
    out.append(f"K{K}g{g}{'w8' if world > 1 else ''} {statistics.median(ts) / K:.3f}")
    f2.close()
print(sys.argv[1], "ms/frame:", "  ".join(out), flush=True)
PY
done; done 2>&1 | tee $O/r04l_fused_counters.log
cp /tmp/keep_l.so $L
