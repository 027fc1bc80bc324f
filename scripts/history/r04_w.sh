#!/bin/bash
# round 4, GPU session W: k_fused_inst with leaves of independent triangles: its test, the instance fuzzer; C4 unchanged
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "fused_pipeline_on_instanced or randomized_instance_fuzz" 2>&1 | tail -3 | tee $O/r04w_pytest.log
timeout 1500 python scripts/fuzz_instances.py 80 9300 2>&1 | tail -6 | tee $O/r04w_fuzz_instances.log
timeout 600 python bench.py --config c4 --pipeline fused --steps 8 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c4 fused', d['value'], d['ms_per_step'])" | tee $O/r04w_c4_fused.log
python - <<'PY' | tee $O/r04w_sah_build_ms.log
import importlib, numpy as np, time
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
for n in (36, 256, 1024, 2047):
    rng = np.random.default_rng(n)
    c = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
    v = (c + rng.uniform(-0.1, 0.1, (n, 3, 3)).astype(np.float32)).reshape(-1)
    f = rng.uniform(0, 1, 6 * n).astype(np.float32)
    for q in (pt.BVH_PREFER_FAST_TRACE, pt.BVH_PREFER_FAST_BUILD):
        t0 = time.perf_counter(); sc = pt.Scene(ctx, v, np.arange(3 * n, dtype=np.uint32), f); t1 = time.perf_counter()
        if q == pt.BVH_PREFER_FAST_BUILD:
            t1 = time.perf_counter(); sc.set_bvh_quality(q)
        t2 = time.perf_counter()
        print(n, "fast_trace" if q == pt.BVH_PREFER_FAST_TRACE else "fast_build", "build_ms", round(sc.info().build_ms, 3), "wall_ms", round((t2 - (t0 if q == pt.BVH_PREFER_FAST_TRACE else t1)) * 1e3, 3))
        sc.close()
PY
