import importlib, numpy as np, time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
n = 2047
rng = np.random.default_rng(n)
c = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
v = (c + rng.uniform(-0.1, 0.1, (n, 3, 3)).astype(np.float32)).reshape(-1)
f = rng.uniform(0, 1, 6 * n).astype(np.float32)
for _ in range(3):
    t0 = time.perf_counter(); sc = pt.Scene(ctx, v, np.arange(3 * n, dtype=np.uint32), f); t1 = time.perf_counter()
    print("build_ms", sc.info().build_ms, "wall", (t1 - t0) * 1e3)
    sc.close()
