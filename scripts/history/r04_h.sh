#!/bin/bash
# round 4, GPU session H: suite (repair test, SAH rows), the SAH rows fixture, fused shards with the centre-first tile order, the
# fused leg, and what the overflow term log is worth to the wavefront C2 shape.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python tests/golden/make_sah_rows.py && cp tests/golden/sah_rows.npz $O/sah_rows.npz
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04h_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r04h_pytest.log
timeout 900 python scripts/probe_shard_efficiency.py 32 fused > $O/r04_shard_efficiency_fused.json 2> $O/r04_shard_efficiency_fused.err; cat $O/r04_shard_efficiency_fused.err
python - <<'PY'
import importlib, json, sys, time
sys.path.insert(0, ".")
import bench
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
scene = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
film = pt.Film(ctx, 1920, 1080)
print("fused leg:", json.dumps(bench.fused_leg(pt, ctx, scene, film, 1920, 1080, 32, 8, 16, 0.0)))
# the wavefront's default shape with and without its overflow term log
kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8, frame=0, frame_count=16)
for knobs in ({}, dict(term_ocap=0), dict(term_ocap=2)):
    old = ctx.set_tuning(**knobs) if knobs else {}
    f2 = pt.Film(ctx, 1920, 1080)
    pt.render(scene, f2, pt.default_params(**kw))
    vals = []
    for _ in range(4):
        f2.clear(); ctx.reset_stats(); t0 = time.perf_counter(); pt.render(scene, f2, pt.default_params(**kw)); vals.append(ctx.stats().rays / (time.perf_counter() - t0) / 1e6)
    st = ctx.stats()
    print("wavefront", knobs or "default", "Mrays/s", [round(v) for v in vals], "redone", st.redone_batches, "groups", st.sample_groups, "ws GB", round(st.workspace_bytes / 2**30, 1))
    f2.close()
    if knobs: ctx.set_tuning(**old)
PY
