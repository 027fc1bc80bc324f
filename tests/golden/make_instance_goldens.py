"""Regenerates tests/golden/instances_independent.npz: an INDEPENDENT evaluation of this build's one extension of the
reference's closest-hit semantics -- instanced scenes (BASELINE config C4; the reference has one identity instance,
main.cpp:515-538, and shades in object space, closesthit.rchit:56-58).

The reference's own compiled shaders (shaders/*.spv, main.cpp:541-543) run in the SPIR-V interpreter oracle/spirv_vm.py as
for tests/golden/spirv_independent.npz, over a driver that shares no code with oracle/ (make_spirv_goldens.IndependentDriver:
binary64 Moeller-Trumbore, numpy sin/cos/sqrt, binary64 dot/cross/normalize, every result rounded once), extended here by
what a Vulkan implementation does for instances and what the build's closest-hit adds on top:
  * traceRayEXT over a TLAS: the ray goes into each instance's object space by the INVERSE of its 3x4 matrix -- numpy's
    binary64 inverse of the binary32 matrix, not the build's "adjugate / determinant rounded once to binary32" -- direction
    not renormalised (t is the same parameter in both spaces), closest t, ties to the lowest (instance, primitive);
  * after the reference's closest-hit shader has filled the payload in OBJECT space, position := M p and
    normal := normalize(M^-T n) in binary64, rounded once ("Instances").
Scene: the Cornell box x 60 rotated, uniformly scaled (0.2 .. 0.6) and translated instances (frozen: seed 4).

Needs /root/reference (build container only); nothing of it is stored: the fixture holds matrices, rays and results.
Usage (repo root): python tests/golden/make_instance_goldens.py        (~2 min on 8 cores)
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_spirv_goldens as G  # noqa: E402  (IndependentDriver, REF, obj_ref, vm)

F32, vm = np.float32, G.vm
N_INST, SEED = 60, 4


def instances():
    rng = np.random.default_rng(SEED)
    m = np.zeros((N_INST, 3, 4), np.float32)
    for k in range(N_INST):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        m[k, :, :3] = (q * rng.uniform(0.2, 0.6)).astype(np.float32)
        m[k, :, 3] = rng.uniform(-1.5, 1.5, 3).astype(np.float32) + np.float32([0, -1, 0])
    return m


class InstancedDriver(G.IndependentDriver):
    def __init__(self, arrays, inst, log=None):
        super().__init__(arrays, log)
        self.M = np.asarray(inst, np.float64)                                   # [k][3][4]
        full = np.concatenate([self.M, np.tile(np.float64([0, 0, 0, 1]), (len(inst), 1, 1))], 1)
        self.Minv = np.linalg.inv(full)[:, :3, :]                                # binary64 inverse
        self.last = None                                                         # (instance, prim) of the last hit

    def trace(self, origin, tmin, direction, tmax):
        o, d = np.array(origin, np.float64), np.array(direction, np.float64)
        oo = self.Minv[:, :, :3] @ o + self.Minv[:, :, 3]                        # [k][3]
        dd = self.Minv[:, :, :3] @ d
        with np.errstate(all="ignore"):
            pv = np.cross(dd[:, None, :], self.e2[None])                         # [k][tri][3]
            det = (self.e1[None] * pv).sum(2)
            tv = oo[:, None, :] - self.v0[None]
            u = (tv * pv).sum(2) / det
            qv = np.cross(tv, self.e1[None])
            vv = (qv * dd[:, None, :]).sum(2) / det
            t = (qv * self.e2[None]).sum(2) / det
        w = 1.0 - u - vv
        inside = (det != 0) & (u >= 0) & (vv >= 0) & (w >= 0) & (t > np.float64(tmin)) & (t < np.float64(tmax))
        flat = np.where(inside, t, np.inf).ravel()
        best = int(np.argmin(flat)) if inside.any() else -1                      # first of equal t = lowest (instance, primitive)
        k, p = (best // t.shape[1], best % t.shape[1]) if best >= 0 else (-1, -1)
        self.last = (k, p)
        if self.log is not None:
            eps = 1e-5
            tb = t[k, p] if best >= 0 else np.inf
            mn = np.minimum(np.minimum(u, vv), w)
            near = (det != 0) & (mn > -eps) & (t > np.float64(tmin) - eps) & (t < tb + eps)
            near_edge = near & (mn < eps)
            near_tmin = near & (np.abs(t - np.float64(tmin)) < eps)
            rivals = near.copy()
            if best >= 0:
                rivals[k] &= ~self.twin[p]                                        # the OBJ's duplicated quads inside the same instance
            clear = not (near_edge.any() or near_tmin.any() or rivals.any())
            self.log.append([*[F32(x) for x in origin], *[F32(x) for x in direction], k, p, F32(tb if best >= 0 else 0),
                             F32(u[k, p]) if best >= 0 else F32(0), F32(vv[k, p]) if best >= 0 else F32(0), clear])
        return None if best < 0 else (p, F32(u[k, p]), F32(vv[k, p]))


class InstancedPipeline(vm.Pipeline):
    """the reference's pipeline + the build's instance-aware closest hit: the payload's object-space position / normal
    go to world space by the instance matrix / its inverse transpose (binary64, one rounding)"""

    def _run_hit_or_miss(self, m, payload_cell, prim=None, attribs=None):
        super()._run_hit_or_miss(m, payload_cell, prim, attribs)
        if m is self.rchit:
            k, _ = self.drv.last
            pay = payload_cell.v
            pos = self.drv.M[k, :, :3] @ np.array(pay[0], np.float64) + self.drv.M[k, :, 3]
            nrm = self.drv.Minv[k, :, :3].T @ np.array(pay[1], np.float64)
            nrm = nrm / np.sqrt((nrm * nrm).sum())
            pay[0] = [F32(x) for x in pos]
            pay[1] = [F32(x) for x in nrm]
            if self.drv.log is not None and self.drv.log:
                self.drv.log[-1] += [*pay[0], *pay[1]]


_st = {}


def run_pixel(job):
    x, y, w, h, want_log = job
    if not _st:
        v, i, f = G.obj_ref.load_obj(os.path.join(G.REPO, "assets", "CornellBox-Original.obj"))
        _st["arrays"] = (v, i, f)
        _st["inst"] = instances()
    log = [] if want_log else None
    drv = InstancedDriver(_st["arrays"], _st["inst"], log)
    pipe = InstancedPipeline(G.REF + "raygen.rgen.spv", G.REF + "closesthit.rchit.spv", G.REF + "miss.rmiss.spv", *_st["arrays"], drv,
                             rgen_int_const_override={32: 1})                    # maxSamples 32 -> 1 (raygen.rgen:43)
    with np.errstate(all="ignore"):
        pipe.launch(x, y, w, h, 0)
    hits = [e for e in (log or []) if e[6] >= 0 and len(e) == 18] + [e + [F32(0)] * 6 for e in (log or []) if e[6] < 0]
    return list(drv.img[(x, y)]), pipe.n_traces, hits


def main():
    t0 = time.time()
    w, h = 120, 68
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        res = pool.map(run_pixel, [(x, y, w, h, (x + y) % 3 == 0) for y in range(h) for x in range(w)], chunksize=32)   # every third pixel's queries are logged
    log = [e for r in res for e in r[2]]
    out = {"launch": np.array([w, h], np.int32), "instances": instances(),
           "texels": np.array([r[0] for r in res], np.float32).reshape(h, w, 4),
           "traces": np.array([r[1] for r in res], np.int32).reshape(h, w),
           "rays6": np.array([e[:6] for e in log], np.float32), "inst": np.array([e[6] for e in log], np.int32),
           "prim": np.array([e[7] for e in log], np.int32), "tuv": np.array([e[8:11] for e in log], np.float32),
           "clear": np.array([e[11] for e in log], bool), "pos": np.array([e[12:15] for e in log], np.float32),
           "nrm": np.array([e[15:18] for e in log], np.float32)}
    np.savez_compressed(os.path.join(HERE, "instances_independent.npz"), **out)
    print(w, "x", h, "1 spp over", N_INST, "instances:", out["traces"].sum(), "traces,", len(log), "logged,",
          int(out["clear"].sum()), "clear, %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
