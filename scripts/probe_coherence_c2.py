"""dev probe: what ray coherence is worth to the LDS (Cornell box) traversal kernel, which is VALU-bound on idle lanes.
Bounce-like rays (origins on the scene's surfaces, uniform hemisphere directions about the normal) traced in random order,
fully sorted by octant (+ cell), and in 512-ray chunks that are only sorted internally by octant (what a block-local
counting sort in k_shade's compaction could deliver for free)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
v, i, f = pt.load_obj(pt.ASSET_CORNELL)
ctx = pt.Context(0); sc = pt.Scene(ctx, v, i, f)
rng = np.random.default_rng(1)
n = 8_000_000
tri = v.reshape(-1, 3, 3)
area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
pick = rng.choice(len(tri), n, p=area / area.sum())
b = rng.dirichlet([1, 1, 1], n).astype(np.float32)
org = (tri[pick] * b[:, :, None]).sum(1).astype(np.float32)
nrm = -np.cross(tri[pick, 1] - tri[pick, 0], tri[pick, 2] - tri[pick, 0]); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
d = np.where(((d * nrm).sum(1) < 0)[:, None], -d, d).astype(np.float32)
rays = np.concatenate([org, d], 1).astype(np.float32)
octant = (d[:, 0] < 0) * 4 + (d[:, 1] < 0) * 2 + (d[:, 2] < 0)
def run(r, label):
    sc.trace(r[:1000])
    ctx.reset_stats(); sc.trace(r); ms = ctx.stats().ms_extend
    print(f"{label:46s} extend {ms:8.3f} ms  {n/ms/1e3:8.1f} Mrays/s")
run(rays, "random order")
run(rays[np.argsort(octant, kind='stable')], "sorted by octant")
q = np.minimum(((org - org.min(0)) / (org.max(0) - org.min(0)) * 4).astype(np.int64), 3)
cell = q[:, 0] * 16 + q[:, 1] * 4 + q[:, 2]
run(rays[np.argsort(cell * 8 + octant, kind='stable')], "sorted by 4^3 cell + octant")
idx = np.arange(n).reshape(-1, 512)
key = octant.reshape(-1, 512)
order = np.take_along_axis(idx, np.argsort(key, axis=1, kind='stable'), axis=1).ravel()
run(rays[order], "512-ray chunks sorted internally by octant")
run(rays[np.argsort(pick, kind='stable')], "sorted by source triangle")
