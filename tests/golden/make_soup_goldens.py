"""Regenerates tests/golden/soup_independent.npz: an INDEPENDENT evaluation of the big-scene path (BASELINE config C5's
recipe at 120 000 triangles: leaves of one triangle, the HBM kernels, the 8-wide tree, materials from the soup's palette).

As for tests/golden/spirv_independent.npz the reference's own compiled shaders (shaders/*.spv, main.cpp:541-543) run in the
SPIR-V interpreter oracle/spirv_vm.py over a driver that shares no code with oracle/ (make_spirv_goldens.IndependentDriver:
traceRayEXT = brute force over ALL triangles with the Moeller-Trumbore test in binary64, closest t, ties to the lowest id;
numpy sin/cos/sqrt, binary64 dot/cross/normalize, every result rounded once).  Fixture:
  * a complete 96x54 launch at 1 sample per pixel (`maxSamples` 32 -> 1, raygen.rgen:43) and the shader's 8 bounces, texels and
    traceRayEXT counts per invocation;
  * every closest-hit query of every third invocation plus 4 000 incoherent rays from inside the soup, with binary64 margins
    ("clear": no edge, tmin plane or rival within 1e-5).
The scene is NOT stored: the soup is the library's frozen generator (host/image_io.cpp pth_make_soup, seed 1), re-made by the
test; the fixture carries a checksum of its arrays.

Needs /root/reference (build container only) and the built libpt_host.so.  Usage (repo root):
    python tests/golden/make_soup_goldens.py        (~3 min on 8 cores)
"""
import hashlib
import importlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_spirv_goldens as G  # noqa: E402  (IndependentDriver, REF, vm)

F32, vm = np.float32, G.vm
N_TRIS, SEED = 120000, 1


def soup():
    sys.path.insert(0, G.REPO)
    pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
    return pt.make_soup(N_TRIS, SEED)


def checksum(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


class _NoTwins:
    """twin[k] of a scene without coincident triangles: only k itself"""

    def __init__(self, n):
        self.n = n

    def __getitem__(self, k):
        m = np.zeros(self.n, bool)
        m[k] = True
        return m


class SoupDriver(G.IndependentDriver):
    """IndependentDriver without the O(n^2) table of coincident triangles (the generator re-draws degenerate ones and
    never repeats a triangle)"""

    def __init__(self, arrays, log=None):
        vm.Driver.__init__(self, lambda a: (F32(np.sin(np.float64(a))), F32(np.cos(np.float64(a)))))
        v, i, _ = arrays
        p = np.asarray(v, np.float64).reshape(-1, 3)[np.asarray(i, np.int64).reshape(-1, 3)]
        self.v0, self.e1, self.e2 = p[:, 0], p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
        self.twin = _NoTwins(len(p))
        self.img, self.log = {}, log


_st = {}


def _arrays():
    if not _st:
        _st["arrays"] = soup()
    return _st["arrays"]


def run_pixel(job):
    x, y, w, h, want_log = job
    arrays = _arrays()
    log = [] if want_log else None
    drv = SoupDriver(arrays, log)
    pipe = vm.Pipeline(G.REF + "raygen.rgen.spv", G.REF + "closesthit.rchit.spv", G.REF + "miss.rmiss.spv", *arrays, drv,
                       rgen_int_const_override={32: 1})                          # maxSamples 32 -> 1 (raygen.rgen:43)
    with np.errstate(all="ignore"):
        pipe.launch(x, y, w, h, 0)
    return list(drv.img[(x, y)]), pipe.n_traces, log or []


def run_rays(rays):
    log = []
    drv = SoupDriver(_arrays(), log)
    with np.errstate(all="ignore"):
        for r in rays:
            drv.trace([F32(c) for c in r[:3]], F32(0.001), [F32(c) for c in r[3:]], F32(10000.0))
    return log


def main():
    t0 = time.time()
    w, h = 96, 54
    arrays = soup()
    rng = np.random.default_rng(5)
    org = (rng.uniform(-0.95, 0.95, (4000, 3)) + np.array([0, -1, 0])).astype(np.float32)
    d = rng.normal(size=(4000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    extra = np.concatenate([org, d], 1)
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        res = pool.map(run_pixel, [(x, y, w, h, (x + y) % 3 == 0) for y in range(h) for x in range(w)], chunksize=16)
        logs = pool.map(run_rays, np.array_split(extra, 32))
    log = [e for r in res for e in r[2]] + [e for part in logs for e in part]
    out = {"launch": np.array([w, h], np.int32), "n_tris": np.int32(N_TRIS), "seed": np.int32(SEED),
           "scene_sha256": np.array(checksum(arrays)),
           "texels": np.array([r[0] for r in res], np.float32).reshape(h, w, 4),
           "traces": np.array([r[1] for r in res], np.int32).reshape(h, w),
           "rays6": np.array([e[:6] for e in log], np.float32), "prim": np.array([e[6] for e in log], np.int32),
           "tuv": np.array([e[7:10] for e in log], np.float32), "clear": np.array([e[10] for e in log], bool)}
    np.savez_compressed(os.path.join(HERE, "soup_independent.npz"), **out)
    print(w, "x", h, "1 spp over", N_TRIS, "triangles:", out["traces"].sum(), "traces,", len(log), "logged queries,",
          int(out["clear"].sum()), "clear,", int((out["prim"] >= 0).sum()), "hits, %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
