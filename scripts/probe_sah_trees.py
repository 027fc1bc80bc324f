"""dev probe: SHA-256 of the BVH4 rows the surface-area builder makes for a set of small scenes (soups of several sizes and spreads, with and without
pair leaves, quads that pair up) and its build time -- run once per library build, the outputs are diffed (scripts/history/r04_ai.sh)."""
import hashlib, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)


def soup(n, seed, spread):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
    v = (c + rng.uniform(-spread, spread, (n, 3, 3)).astype(np.float32)).astype(np.float32)
    return v.reshape(-1), np.arange(3 * n, dtype=np.uint32), rng.uniform(0, 1, 6 * n).astype(np.float32)


def quads(n, seed):   # n quads = 2 n triangles that pair up; a quarter of them coincide in centroid along x (ties)
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    c[: n // 4, 0] = np.float32(0.25)
    e = np.float32([[-0.05, -0.05, 0], [0.05, -0.05, 0], [0.05, 0.05, 0], [-0.05, 0.05, 0]])
    q = c[:, None, :] + e[None]
    t = np.stack([q[:, [0, 1, 2]], q[:, [0, 2, 3]]], 1).reshape(-1, 3, 3)
    return t.reshape(-1).astype(np.float32), np.arange(3 * len(t), dtype=np.uint32), rng.uniform(0, 1, 6 * len(t)).astype(np.float32)


cases = [("soup129", soup(129, 1, 0.1)), ("soup300", soup(300, 2, 0.3)), ("soup700", soup(700, 11, 0.1)), ("soup1500", soup(1500, 3, 0.02)),
         ("soup2047", soup(2047, 4, 0.1)), ("soup2048", soup(2048, 5, 0.5)), ("quads500", quads(500, 6)), ("quads1024", quads(1024, 7)),
         ("cornell", pt.load_obj(pt.ASSET_CORNELL))]
for pair in (1, 0):
    ctx.set_tuning(pair_leaves=pair)
    for name, arrays in cases:
        t0 = time.perf_counter()
        sc = pt.Scene(ctx, *arrays)
        ms = (time.perf_counter() - t0) * 1e3
        assert sc.info().bvh4_builder == 1, name
        rows = sc.read_bvh4()
        print(f"pair_leaves {pair} {name:10s} rows {rows.shape[0]:5d} sha {hashlib.sha256(rows.tobytes()).hexdigest()[:24]}  # {ms:.2f} ms")
        sc.close()
