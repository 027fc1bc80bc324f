"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/pt_oracle.h). The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MISS = 0xFFFFFFFF


class Params(C.Structure):
    _fields_ = [
        ("frame", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32),
        ("spp_per_frame", C.c_uint32), ("max_depth", C.c_uint32),
        ("tmin", C.c_float), ("tmax", C.c_float),
        ("cam_origin", C.c_float * 3), ("cam_target", C.c_float * 3), ("env", C.c_float * 3),
        ("libm_sincos", C.c_uint32), ("nee", C.c_uint32),
    ]


class BvhInfo(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("n_nodes", C.c_uint32), ("height", C.c_uint32),
                ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64)]


HIT_DTYPE = np.dtype([("prim", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("inst", "<u4")])


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build(force=False):
    """gcc -O3 -march=native -ffp-contract=off: the binary is specific to the host CPU, so it is rebuilt when the CPU
    model differs from the one recorded next to it (the in-tree .so travels from the build container to the GPU box)."""
    so = os.path.join(_HERE, "libpt_oracle.so")
    stamp = os.path.join(_HERE, ".cpu_stamp")
    src = [os.path.join(_HERE, f) for f in ("pt_oracle.c", "pt_oracle.h", "Makefile")]
    model = _cpu_model()
    built_on = open(stamp).read().strip() if os.path.exists(stamp) else None
    stale = not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale or built_on != model:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if (force or built_on != model) else []))
        with open(stamp, "w") as fh:
            fh.write(model + "\n")
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        fp = C.POINTER(C.c_float)
        u32p = C.POINTER(C.c_uint32)
        L.orc_pcg.restype = C.c_uint32
        L.orc_pcg.argtypes = [u32p]
        L.orc_pcg2d.argtypes = [C.c_uint32, C.c_uint32, u32p]
        L.orc_rand.restype = C.c_float
        L.orc_rand.argtypes = [u32p]
        L.orc_seed.restype = C.c_uint32
        L.orc_seed.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32]
        L.orc_sincos.argtypes = [C.c_float, fp, fp]
        L.orc_params_default.argtypes = [C.POINTER(Params)]
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_set_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_scene_bvh_info.argtypes = [C.c_void_p, C.POINTER(BvhInfo)]
        L.orc_scene_bvh_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_scene_bvh_nodes.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_trace_batch.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_float,
                                      C.c_float, C.c_void_p, C.POINTER(Counters)]
        L.orc_primary_ray.argtypes = [C.POINTER(Params), C.c_uint32, C.c_uint32, u32p, fp, fp]
        L.orc_shade_hit.argtypes = [C.c_void_p, C.c_void_p, fp, fp, fp, fp]
        L.orc_sample_direction.argtypes = [C.c_float, C.c_float, fp, C.c_int, fp]
        L.orc_render_frame.restype = C.c_uint64
        L.orc_render_frame.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.POINTER(Counters)]
        L.orc_render_rect.restype = C.c_uint64
        L.orc_render_rect.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(Counters)]
        L.orc_render_rect_map.restype = C.c_uint64
        L.orc_render_rect_map.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_void_p]
        L.orc_render_tile_subset.restype = C.c_uint64
        L.orc_render_tile_subset.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p,
                                             C.POINTER(Counters)]
        L.orc_accumulate_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64]
        L.orc_accumulate_bgra8.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64]
        _LIB = L
    return _LIB


def pcg(state):
    s = C.c_uint32(state)
    out = lib().orc_pcg(C.byref(s))
    return s.value, out


def pcg2d(x, y):
    out = (C.c_uint32 * 2)()
    lib().orc_pcg2d(x, y, out)
    return out[0], out[1]


def seed(px, py, sample, frame, spp=32):
    return lib().orc_seed(px, py, sample, frame, spp)


def rands(seed_value, n):
    s = C.c_uint32(seed_value)
    return [float(np.float32(lib().orc_rand(C.byref(s)))) for _ in range(n)]


def sincos(a):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos(C.c_float(a), C.byref(s), C.byref(c))
    return s.value, c.value


def default_params(**kw):
    p = Params()
    lib().orc_params_default(C.byref(p))
    for k, v in kw.items():
        if k in ("cam_origin", "cam_target", "env"):
            setattr(p, k, (C.c_float * 3)(*v))
        else:
            setattr(p, k, v)
    return p


class Scene:
    """The three arrays the reference uploads (main.cpp:492-494)."""

    def __init__(self, vertices, indices, faces):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        self.faces = np.ascontiguousarray(faces, dtype=np.float32).reshape(-1)
        self.n_verts = self.vertices.size // 3
        self.n_tris = self.indices.size // 3
        assert self.faces.size == 6 * self.n_tris
        self.h = lib().orc_scene_create(self.vertices.ctypes.data, self.n_verts,
                                        self.indices.ctypes.data, self.n_tris, self.faces.ctypes.data)
        if not self.h:
            raise ValueError("orc_scene_create failed (bad indices / empty scene)")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_destroy(self.h)
            self.h = None

    def set_instances(self, xforms3x4):
        """n object->world matrices [n,3,4] (VkTransformMatrixKHR layout); empty = single level."""
        x = np.ascontiguousarray(xforms3x4, dtype=np.float32).reshape(-1, 12)
        if lib().orc_scene_set_instances(self.h, x.ctypes.data if len(x) else None, len(x)):
            raise ValueError("orc_scene_set_instances failed")

    def bvh_info(self):
        i = BvhInfo()
        lib().orc_scene_bvh_info(self.h, C.byref(i))
        return i

    def bvh_keys(self):
        keys = np.zeros(self.n_tris, dtype=np.uint64)
        prim = np.zeros(self.n_tris, dtype=np.uint32)
        lib().orc_scene_bvh_keys(self.h, keys.ctypes.data, prim.ctypes.data)
        return keys, prim

    def bvh_nodes(self):
        n = self.bvh_info().n_nodes
        nodes = np.zeros((n, 16), dtype=np.uint32)
        lib().orc_scene_bvh_nodes(self.h, nodes.ctypes.data)
        return nodes

    def trace(self, rays6, tmin=0.001, tmax=10000.0, mode=1):
        rays6 = np.ascontiguousarray(rays6, dtype=np.float32).reshape(-1, 6)
        hits = np.zeros(rays6.shape[0], dtype=HIT_DTYPE)
        cnt = Counters()
        lib().orc_trace_batch(self.h, mode, rays6.shape[0], rays6.ctypes.data, tmin, tmax,
                              hits.ctypes.data, C.byref(cnt))
        return hits, cnt

    def shade_hit(self, hit):
        h = np.zeros(1, dtype=HIT_DTYPE)
        h[0] = hit
        out = [(C.c_float * 3)() for _ in range(4)]
        lib().orc_shade_hit(self.h, h.ctypes.data, *out)
        return [np.array(list(o), dtype=np.float32) for o in out]

    def render_frame(self, params, mode=1, nthreads=None, want_first_hits=False, ray_map=None):
        """-> (frame_color[H,W,3] f32, rays, counters, first_hits or None); ray_map: optional uint32[H,W] that receives the
        number of rays every pixel traced"""
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        w, h = params.width, params.height
        img = np.zeros((h, w, 3), dtype=np.float32)
        fh = np.zeros(h * w, dtype=HIT_DTYPE) if want_first_hits else None
        cnt = Counters()
        if ray_map is not None:
            assert ray_map.dtype == np.uint32 and ray_map.shape == (h, w) and ray_map.flags.c_contiguous
        rays = lib().orc_render_rect_map(self.h, C.byref(params), mode, nthreads, 0, 0, w, h, img.ctypes.data,
                                         fh.ctypes.data if want_first_hits else None, C.byref(cnt),
                                         ray_map.ctypes.data if ray_map is not None else None)
        return img, int(rays), cnt, fh


def render_rect(scene, params, x0, y0, rw, rh, mode=1, nthreads=None):
    """-> (frame_color[rh,rw,3] f32, rays) for the pixel rectangle of the params.width x height launch."""
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    img = np.zeros((rh, rw, 3), dtype=np.float32)
    cnt = Counters()
    rays = lib().orc_render_rect(scene.h, C.byref(params), mode, nthreads, x0, y0, rw, rh, img.ctypes.data, None,
                                 C.byref(cnt))
    return img, int(rays)


def render_tile_subset(scene, params, stride, offset=0, mode=1, nthreads=1):
    """-> (frame_color[H,W,3] f32 with only the 16x16 tiles t % stride == offset rendered, rays)"""
    img = np.zeros((params.height, params.width, 3), dtype=np.float32)
    cnt = Counters()
    rays = lib().orc_render_tile_subset(scene.h, C.byref(params), mode, nthreads, stride, offset, img.ctypes.data, C.byref(cnt))
    return img, int(rays)


def render_tile_subset_counted(scene, params, stride, offset=0, mode=1, nthreads=1):
    """render_tile_subset plus the instrumented walk's counters (binary-LBVH nodes visited, triangles tested) -> (rays, Counters)"""
    img = np.zeros((params.height, params.width, 3), dtype=np.float32)
    cnt = Counters()
    rays = lib().orc_render_tile_subset(scene.h, C.byref(params), mode, nthreads, stride, offset, img.ctypes.data, C.byref(cnt))
    return int(rays), cnt


def primary_ray(params, px, py, seed_value):
    s = C.c_uint32(seed_value)
    o = (C.c_float * 3)()
    d = (C.c_float * 3)()
    lib().orc_primary_ray(C.byref(params), px, py, C.byref(s), o, d)
    return np.array(list(o), np.float32), np.array(list(d), np.float32), s.value


def sample_direction(r1, r2, n, libm=0):
    nn = (C.c_float * 3)(*[float(x) for x in n])
    out = (C.c_float * 3)()
    lib().orc_sample_direction(C.c_float(r1), C.c_float(r2), nn, libm, out)
    return np.array(list(out), np.float32)


def accumulate_f32(film, frame_color, frame):
    assert film.dtype == np.float32 and frame_color.dtype == np.float32
    lib().orc_accumulate_f32(film.ctypes.data, np.ascontiguousarray(frame_color).ctypes.data, frame,
                             film.size // 3)


def accumulate_bgra8(bgra, frame_color, frame):
    assert bgra.dtype == np.uint8 and frame_color.dtype == np.float32
    lib().orc_accumulate_bgra8(bgra.ctypes.data, np.ascontiguousarray(frame_color).ctypes.data, frame,
                               bgra.size // 4)
