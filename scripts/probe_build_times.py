import importlib, numpy as np, time, sys, os
sys.path.insert(0, "/root/repo")
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
arr = pt.load_obj(pt.ASSET_CORNELL)
for n in (0, 300, 1024, 2047):
    if n:
        rng = np.random.default_rng(n)
        c = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
        v = (c + rng.uniform(-0.1, 0.1, (n, 3, 3)).astype(np.float32)).reshape(-1)
        f = rng.uniform(0, 1, 6 * n).astype(np.float32)
        a = (v, np.arange(3 * n, dtype=np.uint32), f)
    else:
        a = arr
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); sc = pt.Scene(ctx, *a); t1 = time.perf_counter()
        ts.append((round(sc.info().build_ms, 2), round((t1 - t0) * 1e3, 2), sc.info().n_wide_nodes)); sc.close()
    print(os.environ.get("PT_LIB_AMD", "product"), n or "cornell", ts, flush=True)
