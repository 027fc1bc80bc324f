#!/bin/bash
# round 4, GPU session AA: the two-launch surface-area builder: build time by scene size (before: 36 3.0 / 256 4.0 / 1024 25.5 / 2047 46.7 ms), tests that build trees
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY' | tee $O/r04aa_sah_build_ms.log
import importlib, numpy as np, time
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
arr = pt.load_obj(pt.ASSET_CORNELL)
for rep in range(3):
    t0 = time.perf_counter(); sc = pt.Scene(ctx, *arr); t1 = time.perf_counter()
    print("cornell build_ms", round(sc.info().build_ms, 3), "wall_ms", round((t1 - t0) * 1e3, 3)); sc.close()
for n in (36, 128, 129, 256, 1024, 2047, 2048):
    rng = np.random.default_rng(n)
    c = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
    v = (c + rng.uniform(-0.1, 0.1, (n, 3, 3)).astype(np.float32)).reshape(-1)
    f = rng.uniform(0, 1, 6 * n).astype(np.float32)
    for rep in range(2):
        t0 = time.perf_counter(); sc = pt.Scene(ctx, v, np.arange(3 * n, dtype=np.uint32), f); t1 = time.perf_counter()
    print(n, "build_ms", round(sc.info().build_ms, 3), "wall_ms", round((t1 - t0) * 1e3, 3))
    sc.close()
PY
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/r04aa_pytest.log
