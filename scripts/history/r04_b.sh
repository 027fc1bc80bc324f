#!/bin/bash
# round 4, GPU session B: the fused pipeline after the batched slot fetch -- its tests, then the headline shape with a refill /
# blocks sweep, K = 2 and K = 1 through the bench's fused leg.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "fused or c3_full" > $O/r04b_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r04b_pytest.log
for t in "" "refill=8" "refill=24" "refill=32" "refill=40" "extend_blocks=4"; do
  PT_TUNE="$t" timeout 300 python bench.py --pipeline fused --no-extra-legs --no-cpu-baseline --reps 3 > $O/r04b_fused_$(echo $t | tr '=' '_').json 2> $O/r04b_fused.err
  python - "$t" $O/r04b_fused_$(echo $t | tr '=' '_').json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("fused", sys.argv[1] or "default", d["value"], d["value_min"], d["value_max"], "ms/frame", d["ms_per_step"], "rays", d["rays"], "groups", d["config"]["sample_groups"], "fif", d["config"]["frames_in_flight"], "ws GB", round(d["workspace_bytes"] / 2**30, 2), "kernel us", d.get("roofline", {}).get("avg_launch_us"))
except Exception as e:
    print("fused", sys.argv[1], "ERR", e, open("gpurun_out/r04b_fused.err").read()[-600:])
PY
done
python - <<'PY'
import importlib, json, sys, time
sys.path.insert(0, ".")
import bench
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
scene = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
film = pt.Film(ctx, 1920, 1080)
print("fused leg:", json.dumps(bench.fused_leg(pt, ctx, scene, film, 1920, 1080, 32, 8, 16, 0.0)))
PY
