"""dev probe: upper bound of what ray sorting can buy on the triangle soup (config C5; `probe_ray_sort.py 8000000` = C5x,
whose traversal working set exceeds the 256 MiB Infinity Cache).
Traces the same incoherent rays (origins on random triangles, uniform hemisphere directions) in random
order and pre-sorted on the host by (origin cell Morton code, direction octant); prints extend-kernel ms."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
v, i, f = pt.make_soup(NT, 1)
ctx = pt.Context(0); sc = pt.Scene(ctx, v, i, f)
rng = np.random.default_rng(1)
n = 4_000_000
tri = v.reshape(-1, 3, 3)[rng.integers(0, NT, n)]
b = rng.dirichlet([1, 1, 1], n).astype(np.float32)
org = (tri * b[:, :, None]).sum(1).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([org, d], 1).astype(np.float32)
def part1by2(x):
    x = x & 0x3FF; x = (x | (x << 16)) & 0x30000FF; x = (x | (x << 8)) & 0x300F00F; x = (x | (x << 4)) & 0x30C30C3; x = (x | (x << 2)) & 0x9249249; return x
def key(bits):
    lo, hi = org.min(0), org.max(0)
    q = np.minimum(((org - lo) / (hi - lo) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    m = (part1by2(q[:, 0]) << 2) | (part1by2(q[:, 1]) << 1) | part1by2(q[:, 2])
    octant = (d[:, 0] < 0).astype(np.int64) * 4 + (d[:, 1] < 0) * 2 + (d[:, 2] < 0)
    return (m << 3) | octant
def run(r, label):
    sc.trace(r[:1000])            # warm
    ctx.reset_stats(); sc.trace(r); ms = ctx.stats().ms_extend
    print(f"{label:40s} extend {ms:8.2f} ms  {n/ms/1e3:8.1f} Mrays/s")
print(f"{NT} triangles, {n} rays, extend variant AUTO")
run(rays, "random order")
for bits in (3, 4, 5, 6, 8):
    run(rays[np.argsort(key(bits), kind='stable')], f"sorted: {bits} bits/axis cell + octant")
run(rays[np.argsort(key(10) >> 3, kind='stable')], "sorted: 10 bits/axis cell only")
