#!/bin/bash
TAG=${1:-r02m}; O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v -E "RCCL|HIP version|ROCm|Hostname|Librccl|amdgpu.ids" | tail -12 ) > $O/${TAG}_pytest.log
python - <<'PY' > $O/${TAG}_sah_time.log 2>&1
import importlib, os, time, numpy as np
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
rng = np.random.default_rng(1)
for n in (36, 300, 1000, 2048):
    if n == 36: v, i, f = pt.load_obj(pt.ASSET_CORNELL)
    else:
        c = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32); v = (c + rng.uniform(-.3, .3, (n, 3, 3)).astype(np.float32)).reshape(-1)
        i = np.arange(3 * n, dtype=np.uint32); f = rng.uniform(0, 1, 6 * n).astype(np.float32)
    for host in ("1", "0"):
        os.environ["PT_TUNE_SAH_HOST"] = host
        t0 = time.perf_counter(); sc = pt.Scene(ctx, v, i, f); ctx.sync(); dt = time.perf_counter() - t0
        print(n, "host" if host == "1" else "device", "scene create wall ms", round(dt * 1e3, 2), "bvh4 nodes", sc.info().n_wide_nodes)
        sc.close()
PY
for i in 1 2; do python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c2', d['value'], r['extend_ms'], r['shade_ms'], r.get('valu_wave_instr_per_64_rays'), r.get('valu_frac'), d['bvh'])"; done >> $O/${TAG}_sah_time.log 2>&1
cat $O/${TAG}_pytest.log $O/${TAG}_sah_time.log
