"""Regenerates tests/golden/spirv_pixels.npz: outputs of the REFERENCE'S OWN compiled shaders
(/root/reference/shaders/{raygen.rgen,closesthit.rchit,miss.rmiss}.spv, the binaries main.cpp:541-543 loads),
executed invocation by invocation by the SPIR-V interpreter `oracle/spirv_vm.py`.

Runs only in the build container (needs /root/reference; nothing of it is copied: the fixture holds launch
sizes, pixel coordinates, frame numbers and the texels / trace counts the shaders produced).

What the interpreter takes from this project instead of from a Vulkan driver -- the parts Vulkan leaves to the
implementation -- is the closest-hit query behind OpTraceRayKHR (`orc_trace`), GLSL.std.450 sin/cos (`orc_sincos`),
correctly rounded sqrt, the evaluation order of dot/cross/normalize, and the unorm8 conversion of the storage
image.  Everything else is the reference's instruction stream.

Usage (repo root):  python tests/golden/make_spirv_goldens.py
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


if __name__ == "__main__":
    main()
