#!/bin/bash
# round 4, GPU session I: the fused kernel's batches scaled by the sample groups -- its tests, the fused leg (K = 16 / 2 / 1), shards.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "fused" > $O/r04i_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r04i_pytest.log
python - <<'PY'
import importlib, json, sys
sys.path.insert(0, ".")
import bench
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
scene = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
film = pt.Film(ctx, 1920, 1080)
print("fused leg:", json.dumps(bench.fused_leg(pt, ctx, scene, film, 1920, 1080, 32, 8, 16, 0.0)))
PY
timeout 900 python scripts/probe_shard_efficiency.py 32 fused > $O/r04i_shard_efficiency_fused.json 2> $O/r04i_shard_efficiency_fused.err; cat $O/r04i_shard_efficiency_fused.err
