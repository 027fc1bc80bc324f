"""Regenerates tests/golden/spirv_pixels.npz: outputs of the REFERENCE'S OWN compiled shaders
(/root/reference/shaders/{raygen.rgen,closesthit.rchit,miss.rmiss}.spv, the binaries main.cpp:541-543 loads),
executed invocation by invocation by the SPIR-V interpreter `oracle/spirv_vm.py`.

Runs only in the build container (needs /root/reference; nothing of it is copied: the fixture holds launch
sizes, pixel coordinates, frame numbers and the texels / trace counts the shaders produced).

What the interpreter takes from this project instead of from a Vulkan driver -- the parts Vulkan leaves to the
implementation -- is the closest-hit query behind OpTraceRayKHR (`orc_trace`), GLSL.std.450 sin/cos (`orc_sincos`),
correctly rounded sqrt, the evaluation order of dot/cross/normalize, and the unorm8 conversion of the storage
image.  Everything else is the reference's instruction stream.

Usage (repo root):  python tests/golden/make_spirv_goldens.py                  (spirv_pixels.npz, canonical driver)
                    python tests/golden/make_spirv_goldens.py --ref1024        (spirv_ref1024.npz: the reference's OWN launch,
                                                                                WIDTH = HEIGHT = 1024, main.cpp:16-17, 659 -- frames 0..3)
                    python tests/golden/make_spirv_goldens.py --independent    (spirv_independent.npz: a driver that shares
                                                                                no code with oracle/, see IndependentDriver)
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import obj_ref  # noqa: E402
from oracle import pt_oracle as O  # noqa: E402
from oracle import spirv_vm as vm  # noqa: E402

REF = "/root/reference/shaders/"
F32 = np.float32


class CanonicalDriver(vm.Driver):
    """the implementation-defined pieces; `unorm8` selects the reference's rgba8 storage image
    (raygen.rgen:7 on B8G8R8A8Unorm memory, main.cpp:483) instead of the float film"""

    def __init__(self, scene, unorm8=False):
        super().__init__(lambda a: O.sincos(float(a)))
        self.scene, self.unorm8, self.img = scene, unorm8, {}

    def trace(self, origin, tmin, direction, tmax):
        h, _ = self.scene.trace(np.array([*origin, *direction], np.float32), tmin=float(tmin), tmax=float(tmax), mode=0)
        return None if h[0]["prim"] == O.MISS else (int(h[0]["prim"]), h[0]["u"], h[0]["v"])

    def image_load(self, x, y):
        t = self.img.get((x, y))
        if t is None:
            return [F32(0)] * 4  # the reference never clears its image; frame 0 multiplies it by 0
        return [F32(b) / F32(255) for b in t] if self.unorm8 else t

    def image_store(self, x, y, texel):
        if self.unorm8:
            q = []
            for c in texel:
                c = F32(c)
                q.append(0 if not c > 0 else int(min(c, F32(1)) * F32(255) + F32(0.5)))
            self.img[(x, y)] = q
        else:
            self.img[(x, y)] = [F32(c) for c in texel]


class IndependentDriver(vm.Driver):
    """A second evaluation of the implementation-defined pieces that shares NO code with oracle/ (pt_oracle.c's
    watertight test, its sin/cos polynomials, its operation order): what an idealised conforming driver could do.
      * traceRayEXT: brute force over all triangles with the Moeller-Trumbore test in binary64 (inputs are the
        shader's binary32 values), tmin < t < tmax, no culling, closest t, equal t -> lowest primitive id; the
        barycentrics are rounded once to binary32.  `log` collects every query with its margins for the hit test.
      * sin / cos / sqrt: numpy binary64, rounded once;  dot / cross / normalize: binary64, rounded once.
    `tests/golden/spirv_independent.npz` is made with it, once with the instruction stream as written and once with
    every multiply-add fused (Pipeline(contract=True))."""

    def __init__(self, arrays, log=None):
        super().__init__(lambda a: (F32(np.sin(np.float64(a))), F32(np.cos(np.float64(a)))))
        v, i, _ = arrays
        p = np.asarray(v, np.float64).reshape(-1, 3)[np.asarray(i, np.int64).reshape(-1, 3)]  # [tri][corner][xyz]
        self.v0, self.e1, self.e2 = p[:, 0], p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
        # triangles with identical corner sets (the OBJ repeats two quads): a tie between them is decided by the id rule
        key = [tuple(sorted(map(tuple, t))) for t in p]
        self.twin = np.array([[key[a] == key[b] for b in range(len(p))] for a in range(len(p))])
        self.img, self.log = {}, log

    def sqrt(self, a): return F32(np.sqrt(np.float64(a)))
    def dot(self, a, b): return F32(sum(np.float64(x) * np.float64(y) for x, y in zip(a, b)))

    def cross(self, a, b):
        a, b = [np.float64(x) for x in a], [np.float64(x) for x in b]
        return [F32(a[1] * b[2] - b[1] * a[2]), F32(a[2] * b[0] - b[2] * a[0]), F32(a[0] * b[1] - b[0] * a[1])]

    def normalize(self, v):
        v = np.array([np.float64(x) for x in v])
        return [F32(x) for x in v / np.sqrt((v * v).sum())]

    def trace(self, origin, tmin, direction, tmax):
        o, d = np.array(origin, np.float64), np.array(direction, np.float64)
        with np.errstate(all="ignore"):
            pv = np.cross(d, self.e2)
            det = (self.e1 * pv).sum(1)
            tv = o - self.v0
            u = (tv * pv).sum(1) / det
            qv = np.cross(tv, self.e1)
            vv = (qv * d).sum(1) / det
            t = (qv * self.e2).sum(1) / det
        w = 1.0 - u - vv
        inside = (det != 0) & (u >= 0) & (vv >= 0) & (w >= 0) & (t > np.float64(tmin)) & (t < np.float64(tmax))
        best = int(np.argmin(np.where(inside, t, np.inf))) if inside.any() else -1  # argmin: first (lowest id) of equal t
        if self.log is not None:
            # clear = no other decision is within 1e-5 of flipping: edges of every triangle that could be the closest
            # hit, the tmin plane, and the runner-up's distance (identical twins excepted)
            eps = 1e-5
            tb = t[best] if best >= 0 else np.inf
            near = (det != 0) & (np.minimum(np.minimum(u, vv), w) > -eps) & (t > np.float64(tmin) - eps) & (t < tb + eps)
            near_edge = near & (np.minimum(np.minimum(u, vv), w) < eps)
            near_tmin = near & (np.abs(t - np.float64(tmin)) < eps)
            rivals = near.copy()
            if best >= 0:
                rivals &= ~self.twin[best]
            clear = not (near_edge.any() or near_tmin.any() or rivals.any())
            self.log.append((*[F32(x) for x in origin], *[F32(x) for x in direction], best, F32(tb if best >= 0 else 0),
                             F32(u[best]) if best >= 0 else F32(0), F32(vv[best]) if best >= 0 else F32(0), clear))
        return None if best < 0 else (best, F32(u[best]), F32(vv[best]))

    def image_load(self, x, y):
        return self.img.get((x, y), [F32(0)] * 4)

    def image_store(self, x, y, texel):
        self.img[(x, y)] = [F32(c) for c in texel]


_state = {}


def _init():
    v, i, f = obj_ref.load_obj(os.path.join(REPO, "assets", "CornellBox-Original.obj"))
    _state["arrays"] = (v, i, f)
    _state["scene"] = O.Scene(v, i, f)


def run_pixel(job):
    """job = (x, y, width, height, n_frames, unorm8) -> (texel after each frame [n,4], traces per frame [n])"""
    x, y, w, h, n_frames, unorm8 = job
    if not _state:
        _init()
    drv = CanonicalDriver(_state["scene"], unorm8)
    pipe = vm.Pipeline(REF + "raygen.rgen.spv", REF + "closesthit.rchit.spv", REF + "miss.rmiss.spv", *_state["arrays"], drv)
    tex, rays = [], []
    with np.errstate(all="ignore"):
        for frame in range(n_frames):
            n0 = pipe.n_traces
            pipe.launch(x, y, w, h, frame)
            tex.append(list(drv.img[(x, y)]))
            rays.append(pipe.n_traces - n0)
    return tex, rays


def run_pixel_independent(job):
    """job = (x, y, width, height, n_frames, spp (None = the shader's 32), contract, log rays?) ->
    (texel after each frame [n,4], traces per frame [n], ray log)"""
    x, y, w, h, n_frames, spp, contract, want_log = job
    if not _state:
        _init()
    log = [] if want_log else None
    drv = IndependentDriver(_state["arrays"], log)
    pipe = vm.Pipeline(REF + "raygen.rgen.spv", REF + "closesthit.rchit.spv", REF + "miss.rmiss.spv", *_state["arrays"], drv,
                       contract=contract, rgen_int_const_override=None if spp is None else {32: spp})
    tex, rays = [], []
    with np.errstate(all="ignore"):
        for frame in range(n_frames):
            n0 = pipe.n_traces
            pipe.launch(x, y, w, h, frame)
            tex.append(list(drv.img[(x, y)]))
            rays.append(pipe.n_traces - n0)
    return tex, rays, log


def main_independent():
    """tests/golden/spirv_independent.npz: the reference's shaders over the INDEPENDENT driver (no oracle code)."""
    t0 = time.time()
    out = {}
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        # E: a complete 240x136 launch at ONE sample per pixel (maxSamples specialised 32 -> 1), frame 0
        w, h = 240, 136
        for name, contract in (("ideal", False), ("fma", True)):
            res = pool.map(run_pixel_independent, [(x, y, w, h, 1, 1, contract, name == "ideal" and x % 4 == 0 and y % 2 == 0)
                                                   for y in range(h) for x in range(w)], chunksize=64)
            out["e_texels_" + name] = np.array([r[0][0] for r in res], np.float32).reshape(h, w, 4)
            out["e_traces_" + name] = np.array([r[1][0] for r in res], np.int32).reshape(h, w)
            if name == "ideal":
                log = [e for r in res if r[2] for e in r[2]]
                out["e_rays6"] = np.array([e[:6] for e in log], np.float32)
                out["e_prim"] = np.array([e[6] for e in log], np.int32)
                out["e_tuv"] = np.array([e[7:10] for e in log], np.float32)
                out["e_clear"] = np.array([e[10] for e in log], bool)
            print("E", name, w, "x", h, "1 spp, traces", out["e_traces_" + name].sum(), "%.0f s" % (time.time() - t0), flush=True)
        out["e_launch"] = np.array([w, h], np.int32)
        # F: the launch of fixture C (120x68, the shader's own 32 spp, frames 0 and 1 = 64 spp)
        w, h = 120, 68
        for name, contract in (("ideal", False), ("fma", True)):
            res = pool.map(run_pixel_independent, [(x, y, w, h, 2, None, contract, False) for y in range(h) for x in range(w)], chunksize=8)
            out["f_texels_" + name] = np.array([r[0] for r in res], np.float32).reshape(h, w, 2, 4).transpose(2, 0, 1, 3)
            out["f_traces_" + name] = np.array([r[1] for r in res], np.int64).reshape(h, w, 2).transpose(2, 0, 1)
            print("F", name, "traces", out["f_traces_" + name].sum(axis=(1, 2)), "%.0f s" % (time.time() - t0), flush=True)
        out["f_launch"] = np.array([w, h], np.int32)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "spirv_independent.npz"), **out)


def pixel_set(w, h, nx, ny, seed):
    rng = np.random.default_rng(seed)
    px = [(0, 0), (1, 0), (0, 1), (w - 1, h - 1), (w - 1, 0), (0, h - 1), (w // 2, h // 2), (5, 0), (0, 7), (w // 2, 0), (0, h // 2)]
    for j in range(ny):
        for i in range(nx):
            px.append((int((i + rng.random()) * w / nx), int((j + rng.random()) * h / ny)))
    # the emitter (primary rays that end on the light) and the box silhouettes around the image centre
    for _ in range(24):
        px.append((int(w * (0.42 + 0.16 * rng.random())), int(h * (0.02 + 0.10 * rng.random()))))
    return np.array(sorted(set(px)), dtype=np.int32)


def main():
    t0 = time.time()
    out = {}
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        # A: BASELINE configs 2/3 launch size, progressive frames 0..2 into a float image
        W, H, NF = 1920, 1080, 3
        px = pixel_set(W, H, 20, 12, 1)
        res = pool.map(run_pixel, [(int(x), int(y), W, H, NF, False) for x, y in px], chunksize=4)
        out["a_launch"] = np.array([W, H], np.int32)
        out["a_pixels"] = px
        out["a_texels"] = np.array([r[0] for r in res], np.float32).transpose(1, 0, 2)  # [frame][pixel][rgba]
        out["a_traces"] = np.array([r[1] for r in res], np.int64).T
        print("A", len(px), "pixels x", NF, "frames, traces", out["a_traces"].sum(), "%.0f s" % (time.time() - t0))
        # B: the same launch through the reference's 8-bit storage image, frames 0..3
        pxb = px[::6]
        res = pool.map(run_pixel, [(int(x), int(y), W, H, 4, True) for x, y in pxb], chunksize=2)
        out["b_pixels"] = pxb
        out["b_rgba8"] = np.array([r[0] for r in res], np.uint8).transpose(1, 0, 2)  # component order r,g,b,a
        print("B", len(pxb), "pixels x 4 frames %.0f s" % (time.time() - t0))
        # C: one complete small launch (ragged size), frames 0..1
        w, h = 120, 68
        res = pool.map(run_pixel, [(x, y, w, h, 2, False) for y in range(h) for x in range(w)], chunksize=8)
        out["c_launch"] = np.array([w, h], np.int32)
        out["c_texels"] = np.array([r[0] for r in res], np.float32).reshape(h, w, 2, 4).transpose(2, 0, 1, 3)
        out["c_traces"] = np.array([r[1] for r in res], np.int64).reshape(h, w, 2).transpose(2, 0, 1)
        print("C", w, "x", h, "x 2 frames, traces", out["c_traces"].sum(axis=(1, 2)), "%.0f s" % (time.time() - t0))
        # D: the rectangle of tests/golden/c2_crop_1080p_32spp_d8.npz (that file came from the oracle; this one from
        # the reference's shaders), frames 0..1
        # (progressive: texel[1] is the running mean of frames 0 and 1)
        x0, y0, rw, rh = 912, 508, 96, 64
        res = pool.map(run_pixel, [(x0 + x, y0 + y, W, H, 2, False) for y in range(rh) for x in range(rw)], chunksize=16)
        out["d_rect"] = np.array([x0, y0, rw, rh], np.int32)
        out["d_texels"] = np.array([r[0] for r in res], np.float32).reshape(rh, rw, 2, 4).transpose(2, 0, 1, 3)
        out["d_traces"] = np.array([r[1] for r in res], np.int64).reshape(rh, rw, 2).transpose(2, 0, 1)
        print("D crop", rw, "x", rh, "traces", out["d_traces"].sum(axis=(1, 2)), "%.0f s" % (time.time() - t0))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "spirv_pixels.npz"), **out)


def main_ref1024():
    """tests/golden/spirv_ref1024.npz: the reference's own dispatch -- traceRaysKHR(1024, 1024, 1) (main.cpp:16-17, 659), push constant frame = 0, 1, 2,
    3 (main.cpp:656-658), the image blended in place (raygen.rgen:88-90) -- executed by the reference's compiled shaders for a 64 x 48 rectangle that holds the
    tall box's front, its coincident duplicate and the light's reflection, and for a scattered pixel set (corners, row / column 0, the emitter, a jittered
    grid); a sixth of the scattered set also through the rgba8 storage image."""
    t0 = time.time()
    out = {}
    W = H = 1024
    NF = 4
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        x0, y0, rw, rh = 480, 488, 64, 48
        res = pool.map(run_pixel, [(x0 + x, y0 + y, W, H, NF, False) for y in range(rh) for x in range(rw)], chunksize=16)
        out["launch"] = np.array([W, H], np.int32)
        out["rect"] = np.array([x0, y0, rw, rh], np.int32)
        out["rect_texels"] = np.array([r[0] for r in res], np.float32).reshape(rh, rw, NF, 4).transpose(2, 0, 1, 3)   # [frame][y][x][rgba]
        out["rect_traces"] = np.array([r[1] for r in res], np.int64).reshape(rh, rw, NF).transpose(2, 0, 1)
        print("ref1024 rect", rw, "x", rh, "x", NF, "frames, traces", out["rect_traces"].sum(axis=(1, 2)), "%.0f s" % (time.time() - t0), flush=True)
        px = pixel_set(W, H, 16, 16, 3)
        res = pool.map(run_pixel, [(int(x), int(y), W, H, NF, False) for x, y in px], chunksize=4)
        out["pixels"] = px
        out["texels"] = np.array([r[0] for r in res], np.float32).transpose(1, 0, 2)    # [frame][pixel][rgba]
        out["traces"] = np.array([r[1] for r in res], np.int64).T
        print("ref1024", len(px), "pixels x", NF, "frames, traces", out["traces"].sum(), "%.0f s" % (time.time() - t0), flush=True)
        pxb = px[::6]
        res = pool.map(run_pixel, [(int(x), int(y), W, H, NF, True) for x, y in pxb], chunksize=2)
        out["pixels_rgba8"] = pxb
        out["rgba8"] = np.array([r[0] for r in res], np.uint8).transpose(1, 0, 2)        # component order r,g,b,a
        print("ref1024 rgba8", len(pxb), "pixels %.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "spirv_ref1024.npz"), **out)


if __name__ == "__main__":
    if "--ref1024" in sys.argv:
        main_ref1024()
    elif "--independent" in sys.argv:
        main_independent()
    elif "--all" in sys.argv:
        main()
        main_independent()
    else:
        main()
