"""MI355X-native wavefront path tracer -- Python mirror of the C-ABI (include/pt_api.h, pt_host.h).

The package directory name contains '-', so import it with
    importlib.import_module("single-file-vulkan-pathtracing_amd")
(`__graft_entry__.py`, `bench.py` and the tests do exactly that).

This module only loads the in-tree shared libraries and forwards to them:
    libpt_amd.so   hand-written HIP kernels for gfx950 + the pt_* entry points
    libpt_host.so  C++20 host code: OBJ/MTL ingest (reference loadFromFile, main.cpp:28-58), image output
There is NO CPU fallback: without the HIP library, or without a GPU, every compute call raises.
Names follow the reference: Scene = vertex/index/face buffers + acceleration structure
(main.cpp:492-538), Film = the storage image (main.cpp:481-484), render() = pushConstants +
traceRaysKHR (main.cpp:656-659).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
ASSET_CORNELL = os.path.join(REPO, "assets", "CornellBox-Original.obj")

PT_OK = 0
STATUS_NAMES = {0: "PT_OK", 1: "PT_ERR_INVALID_ARG", 2: "PT_ERR_NO_DEVICE", 3: "PT_ERR_HIP", 4: "PT_ERR_OOM",
                5: "PT_ERR_UNSUPPORTED"}
PIPELINE_WAVEFRONT = 0
PIPELINE_WAVEFRONT_NEE = 1
PIPELINE_FUSED = 2
PIPELINE_AUTO = 3
PIPELINE_NAMES = {0: "wavefront", 1: "wavefront_nee", 2: "fused", 3: "auto"}
FLAG_PROFILE = 1
FLAG_COUNT_VISITS = 2
FLAG_ASYNC = 4
FLAG_SORT_RAYS = 8
FLAG_NO_SORT_RAYS = 16
EXTEND_AUTO, EXTEND_LDS, EXTEND_HBM, EXTEND_HBM8 = 0, 2, 3, 4
EXTEND_FLAT = 1   # deprecated: the brute-force loop was removed in API version 5 (PT_ERR_UNSUPPORTED); the name is kept for source compatibility
BVH_PREFER_FAST_TRACE, BVH_PREFER_FAST_BUILD = 0, 1
EXTEND_NAMES = {2: "BVH4, scene staged in LDS", 3: "BVH4, scene in HBM/L2",
                4: "8-wide tree (64-B nodes with byte planes, one stack entry per node), scene in HBM/L2"}
MISS = 0xFFFFFFFF


class PtError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status


class Params(C.Structure):
    _fields_ = [
        ("frame", C.c_int32), ("frame_count", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
        ("spp_per_frame", C.c_uint32), ("max_depth", C.c_uint32), ("tmin", C.c_float), ("tmax", C.c_float),
        ("cam_origin", C.c_float * 3), ("cam_target", C.c_float * 3), ("env", C.c_float * 3),
        ("rank", C.c_uint32), ("world", C.c_uint32), ("pipeline", C.c_uint32),
        ("frames_in_flight", C.c_uint32), ("flags", C.c_uint32), ("extend", C.c_uint32),
        ("sample_groups", C.c_uint32),
    ]


class SceneInfo(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("n_nodes", C.c_uint32), ("bvh_height", C.c_uint32),
                ("n_wide_nodes", C.c_uint32), ("n_instances", C.c_uint32), ("n_tlas_nodes", C.c_uint32),
                ("leaf_max", C.c_uint32), ("bvh4_builder", C.c_uint32),
                ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3), ("build_ms", C.c_float),
                ("device_bytes", C.c_uint64), ("n_wide8_nodes", C.c_uint32), ("wide8_levels", C.c_uint32),
                ("device_bytes8", C.c_uint64), ("tree_area_lbvh", C.c_float), ("tree_area_ploc", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("rounds", C.c_uint32),
                ("launches_extend", C.c_uint32), ("launches_shade", C.c_uint32), ("launches_other", C.c_uint32),
                ("ms_total", C.c_float), ("ms_extend", C.c_float), ("ms_shade", C.c_float),
                ("extend_variant", C.c_uint32), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("frames_in_flight", C.c_uint32), ("sample_groups", C.c_uint32),
                ("node_steps", C.c_uint64), ("tri_steps", C.c_uint64),
                ("redone_batches", C.c_uint32), ("pipelines", C.c_uint32),
                ("wave_refills", C.c_uint64), ("wave_pops", C.c_uint64), ("wave_hit_blocks", C.c_uint64),
                ("wave_finishes", C.c_uint64), ("wave_iterations", C.c_uint64),
                ("leaf_lanes", C.c_uint64), ("pop_lanes", C.c_uint64), ("hit_lanes", C.c_uint64),
                ("enter_steps", C.c_uint64), ("enter_lanes", C.c_uint64), ("workspace_bytes", C.c_uint64),
                ("pipeline", C.c_uint32), ("tail_samples", C.c_uint32), ("rays_culled", C.c_uint64)]


TUNING_NAMES = ["refill", "lds_stack", "extend_blocks", "pipes", "stagger", "sort_bits", "pair_leaves", "pair_kernel", "topdown4",
                "rec64", "inst16", "inst16_blocks", "enter_min", "node_yield", "tlas_lds_kb", "term_ocap", "term_spill", "mem_budget_mb",
                "hbm8", "ploc_radius", "leaf_min", "tri_enter", "tri_stay", "inst_frames", "tlas_ploc", "ploc_adopt_pct", "fail_rebuild", "fused_tail", "fused_subject", "cull"]


class Tuning(C.Structure):
    """include/pt_api.h pt_tuning: speed knobs of a context, -1 = the built-in choice; never changes a result."""
    _fields_ = [(n, C.c_int32) for n in TUNING_NAMES] + [("reserved", C.c_int32 * 2)]


class HostScene(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("n_verts", C.c_uint32), ("indices", C.POINTER(C.c_uint32)),
                ("n_tris", C.c_uint32), ("faces", C.POINTER(C.c_float))]


# include/pt_api.h pt_fused_block, in order
FUSED_BLOCKS = ["ITER", "SHADE", "HIT", "MISS", "SURFACE", "ADD", "BOUNCE", "NEXT", "DONE", "HANDOUT", "DRAW", "TAKE", "CULLED", "PRIMARY", "SETUP",
                "NODE", "POP", "LEAF", "DIV", "FINISH", "TRACE", "SPAWN", "PTARGET", "PDIR", "POPTOP"]
HIT_DTYPE = np.dtype([("prim", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("inst", "<u4")])

# every symbol include/pt_api.h and include/pt_host.h declare
API_SYMBOLS = ["pt_ctx_create", "pt_ctx_destroy", "pt_last_error", "pt_sync", "pt_scene_create", "pt_scene_destroy",
               "pt_scene_set_instances", "pt_scene_set_bvh_quality",
               "pt_scene_get_info", "pt_scene_read_bvh", "pt_scene_read_bvh4", "pt_scene_read_bvh8", "pt_film_create", "pt_film_create_external", "pt_film_clear",
               "pt_film_read_f32", "pt_film_read_bgra8", "pt_film_destroy", "pt_params_default", "pt_render",
               "pt_render_prepare", "pt_trace",
               "pt_get_stats", "pt_reset_stats", "pt_get_block_counts",
               "pt_comm_unique_id", "pt_comm_create", "pt_comm_ranks", "pt_comm_destroy", "pt_film_present",
               "pt_film_tile_count", "pt_film_pack_tiles", "pt_film_unpack_tiles",
               "pt_device_alloc", "pt_device_free", "pt_device_read", "pt_device_write", "pt_ctx_get_tuning", "pt_ctx_set_tuning"]
HOST_SYMBOLS = ["pth_load_obj", "pth_load_obj_ex", "pth_free_scene", "pth_write_ppm_bgra8", "pth_write_pfm", "pth_write_soup_obj", "pth_make_soup",
                "pth_make_stadium"]

_amd = None
_host = None


def build(verbose=False):
    """Compile the HIP library for gfx950 and the host library/driver, in-tree."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], stdout=out)
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "host")], stdout=out)


def lib_amd():
    global _amd
    if _amd is None:
        # PT_LIB_AMD: a development build of the same library (scripts/: timeline instrumentation, an older revision for an A/B)
        path = os.environ.get("PT_LIB_AMD") or os.path.join(_HERE, "libpt_amd.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                              "there is no CPU fallback")
        L = C.CDLL(path)
        vp = C.c_void_p
        L.pt_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
        L.pt_ctx_destroy.argtypes = [vp]
        L.pt_ctx_destroy.restype = None
        L.pt_last_error.argtypes = [vp]
        L.pt_last_error.restype = C.c_char_p
        L.pt_sync.argtypes = [vp]
        L.pt_ctx_get_tuning.argtypes = [vp, C.POINTER(Tuning)]
        L.pt_ctx_set_tuning.argtypes = [vp, C.POINTER(Tuning)]
        L.pt_scene_create.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.POINTER(vp)]
        L.pt_scene_destroy.argtypes = [vp]
        L.pt_scene_destroy.restype = None
        L.pt_scene_set_instances.argtypes = [vp, vp, C.c_uint32]
        L.pt_scene_set_bvh_quality.argtypes = [vp, C.c_uint32]
        L.pt_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
        L.pt_scene_read_bvh.argtypes = [vp, vp, vp, vp]
        L.pt_scene_read_bvh4.argtypes = [vp, vp]
        L.pt_scene_read_bvh8.argtypes = [vp, vp, vp]
        L.pt_film_create.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.pt_film_create_external.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.POINTER(vp)]
        L.pt_film_clear.argtypes = [vp]
        L.pt_film_read_f32.argtypes = [vp, vp]
        L.pt_film_read_bgra8.argtypes = [vp, vp]
        L.pt_film_destroy.argtypes = [vp]
        L.pt_film_destroy.restype = None
        L.pt_params_default.argtypes = [C.POINTER(Params)]
        L.pt_params_default.restype = None
        L.pt_render.argtypes = [vp, vp, C.POINTER(Params)]
        L.pt_render_prepare.argtypes = [vp, vp, C.POINTER(Params)]
        L.pt_trace.argtypes = [vp, vp, C.c_uint32, C.c_float, C.c_float, C.c_uint32, vp]
        L.pt_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.pt_reset_stats.argtypes = [vp]
        if hasattr(L, "pt_get_block_counts"):   # (API version 6; dev A/B runs load older builds through PT_LIB_AMD)
            L.pt_get_block_counts.argtypes = [vp, C.POINTER(C.c_uint64), C.c_uint32]
        L.pt_comm_unique_id.argtypes = [vp]
        L.pt_comm_create.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.pt_comm_ranks.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.pt_comm_destroy.argtypes = [vp]
        L.pt_comm_destroy.restype = None
        L.pt_film_present.argtypes = [vp, vp, C.c_uint32, vp]
        L.pt_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.pt_device_free.argtypes = [vp, vp]
        L.pt_device_read.argtypes = [vp, vp, vp, C.c_size_t]
        L.pt_device_write.argtypes = [vp, vp, vp, C.c_size_t]
        L.pt_film_tile_count.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.pt_film_pack_tiles.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
        L.pt_film_unpack_tiles.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp]
        _amd = L
    return _amd


def lib_host():
    global _host
    if _host is None:
        path = os.path.join(_HERE, "libpt_host.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run __graft_entry__.build()")
        L = C.CDLL(path)
        L.pth_load_obj.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(HostScene), C.c_char_p, C.c_size_t]
        L.pth_load_obj_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(HostScene), C.c_char_p, C.c_size_t]
        L.pth_free_scene.argtypes = [C.POINTER(HostScene)]
        L.pth_free_scene.restype = None
        L.pth_write_ppm_bgra8.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.pth_write_pfm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.pth_write_soup_obj.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
        L.pth_make_soup.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(HostScene)]
        L.pth_make_stadium.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(HostScene)]
        _host = L
    return _host


# ---- host side: scene ingest / image output ---------------------------------------------------
QUAD_SHORTER_DIAGONAL = 1
SMALL_CHUNKS = 2  # test hook of the loader (include/pt_host.h)


def load_obj(path, mtl_dir=None, flags=0):
    """-> (vertices f32[3*nv], indices u32[3*nt], faces f32[6*nt]) as the reference's loadFromFile fills them
    (main.cpp:28-58); flags: QUAD_SHORTER_DIAGONAL (include/pt_host.h)."""
    hs = HostScene()
    err = C.create_string_buffer(512)
    rc = lib_host().pth_load_obj_ex(os.fsencode(path), os.fsencode(mtl_dir) if mtl_dir else None, flags, C.byref(hs), err, 512)
    if rc != 0:
        raise RuntimeError(f"load_obj({path}): {err.value.decode()}")  # reference: throw std::runtime_error (main.cpp:35)
    try:
        v = np.ctypeslib.as_array(hs.vertices, shape=(3 * hs.n_verts,)).copy()
        i = np.ctypeslib.as_array(hs.indices, shape=(3 * hs.n_tris,)).copy()
        f = np.ctypeslib.as_array(hs.faces, shape=(6 * hs.n_tris,)).copy()
    finally:
        lib_host().pth_free_scene(C.byref(hs))
    return v, i, f


def write_ppm(path, bgra):
    h, w = bgra.shape[:2]
    a = np.ascontiguousarray(bgra, dtype=np.uint8)
    if lib_host().pth_write_ppm_bgra8(os.fsencode(path), a.ctypes.data, w, h) != 0:
        raise RuntimeError(f"cannot write {path}")


def write_pfm(path, rgb):
    h, w = rgb.shape[:2]
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    if lib_host().pth_write_pfm(os.fsencode(path), a.ctypes.data, w, h) != 0:
        raise RuntimeError(f"cannot write {path}")


def write_soup_obj(path, n_tris, seed=1):
    if lib_host().pth_write_soup_obj(os.fsencode(path), n_tris, seed) != 0:
        raise RuntimeError(f"cannot write {path}")


def make_soup(n_tris, seed=1):
    """-> the arrays load_obj(write_soup_obj(...)) would return, without the OBJ text in between."""
    hs = HostScene()
    if lib_host().pth_make_soup(n_tris, seed, C.byref(hs)) != 0:
        raise RuntimeError("pth_make_soup failed")
    try:
        v = np.ctypeslib.as_array(hs.vertices, shape=(3 * hs.n_verts,)).copy()
        i = np.ctypeslib.as_array(hs.indices, shape=(3 * hs.n_tris,)).copy()
        f = np.ctypeslib.as_array(hs.faces, shape=(6 * hs.n_tris,)).copy()
    finally:
        lib_host().pth_free_scene(C.byref(hs))
    return v, i, f


def make_stadium(floor_side=384, sphere_seg=160):
    """The "teapot in a stadium" stress scene (include/pt_host.h pth_make_stadium): -> (vertices, indices, faces)."""
    hs = HostScene()
    if lib_host().pth_make_stadium(floor_side, sphere_seg, C.byref(hs)) != 0:
        raise RuntimeError("pth_make_stadium failed")
    try:
        v = np.ctypeslib.as_array(hs.vertices, shape=(3 * hs.n_verts,)).copy()
        i = np.ctypeslib.as_array(hs.indices, shape=(3 * hs.n_tris,)).copy()
        f = np.ctypeslib.as_array(hs.faces, shape=(6 * hs.n_tris,)).copy()
    finally:
        lib_host().pth_free_scene(C.byref(hs))
    return v, i, f


def cornell_grid_instances(n=100, cell=0.02, scale=0.009):
    """BASELINE config C4 (frozen recipe): n x n instances of the scene in the
    z = 0 plane filling x in [-1,1], y in [-2,0]: uniform scale `scale`, translation
    (-1 + (i+1/2) cell, -2 + (j+1/2) cell + scale, 0); row-major 3x4, instance id = j*n + i."""
    m = np.zeros((n * n, 3, 4), dtype=np.float32)
    i = np.arange(n, dtype=np.float32)
    tx = (np.float32(-1.0) + (i + np.float32(0.5)) * np.float32(cell)).astype(np.float32)
    ty = (np.float32(-2.0) + (i + np.float32(0.5)) * np.float32(cell) + np.float32(scale)).astype(np.float32)
    m[:, 0, 0] = m[:, 1, 1] = m[:, 2, 2] = np.float32(scale)
    m[:, 0, 3] = np.tile(tx, n)
    m[:, 1, 3] = np.repeat(ty, n)
    return m


# ---- device side --------------------------------------------------------------------------------
def default_params(**kw):
    p = Params()
    lib_amd().pt_params_default(C.byref(p))
    for k, v in kw.items():
        if k in ("cam_origin", "cam_target", "env"):
            setattr(p, k, (C.c_float * 3)(*v))
        else:
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
    return p


class Context:
    """One GPU + one stream (reference: Context, main.cpp:74-267)."""

    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        rc = lib_amd().pt_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self.h))
        if rc != PT_OK:
            raise PtError(rc, lib_amd().pt_last_error(None).decode())

    def _check(self, rc):
        if rc != PT_OK:
            raise PtError(rc, lib_amd().pt_last_error(self.h).decode())

    def sync(self):
        self._check(lib_amd().pt_sync(self.h))

    def tuning(self):
        t = Tuning()
        self._check(lib_amd().pt_ctx_get_tuning(self.h, C.byref(t)))
        return t

    def set_tuning(self, **kw):
        """Change speed knobs (names: TUNING_NAMES; -1 = built-in choice) -> the previous values of the ones changed, so a
        caller can put them back: ctx.set_tuning(**ctx.set_tuning(node_yield=0))."""
        t = self.tuning()
        old = {}
        for k, v in kw.items():
            if k not in TUNING_NAMES:
                raise AttributeError(k)
            old[k] = getattr(t, k)
            setattr(t, k, int(v))
        self._check(lib_amd().pt_ctx_set_tuning(self.h, C.byref(t)))
        return old

    def stats(self):
        s = Stats()
        self._check(lib_amd().pt_get_stats(self.h, C.byref(s)))
        return s

    def reset_stats(self):
        self._check(lib_amd().pt_reset_stats(self.h))

    def block_counts(self):
        """{block: (wave executions, lanes inside)} of the instrumented fused kernel since reset_stats (pt_get_block_counts; render with
        pipeline=PIPELINE_FUSED, flags=FLAG_COUNT_VISITS)."""
        a = (C.c_uint64 * (2 * len(FUSED_BLOCKS)))()
        self._check(lib_amd().pt_get_block_counts(self.h, a, len(FUSED_BLOCKS)))
        return {n: (int(a[2 * i]), int(a[2 * i + 1])) for i, n in enumerate(FUSED_BLOCKS)}

    def close(self):
        if self.h:
            lib_amd().pt_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """Vertex/index/face buffers + the device-built LBVH (reference: main.cpp:492-538)."""

    def __init__(self, ctx, vertices, indices, faces):
        self.ctx = ctx
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        f = np.ascontiguousarray(faces, dtype=np.float32).reshape(-1)
        if v.size % 3 or i.size % 3 or f.size != 2 * i.size:
            raise ValueError("vertices must be 3*nv, indices 3*nt, faces 6*nt")
        self.h = C.c_void_p()
        ctx._check(lib_amd().pt_scene_create(ctx.h, v.ctypes.data, v.size // 3, i.ctypes.data, i.size // 3,
                                             f.ctypes.data, C.byref(self.h)))

    @classmethod
    def from_obj(cls, ctx, path=ASSET_CORNELL):
        return cls(ctx, *load_obj(path))

    def set_instances(self, xforms3x4):
        """n object->world 3x4 matrices (VkTransformMatrixKHR layout); empty = the reference's
        single identity instance (main.cpp:515-538)."""
        x = np.ascontiguousarray(xforms3x4, dtype=np.float32).reshape(-1, 12)
        self.ctx._check(lib_amd().pt_scene_set_instances(self.h, x.ctypes.data if len(x) else None, len(x)))

    def set_bvh_quality(self, quality):
        """BVH_PREFER_FAST_TRACE (default; main.cpp:419) or BVH_PREFER_FAST_BUILD (always the collapsed LBVH)."""
        self.ctx._check(lib_amd().pt_scene_set_bvh_quality(self.h, quality))

    def info(self):
        i = SceneInfo()
        self.ctx._check(lib_amd().pt_scene_get_info(self.h, C.byref(i)))
        return i

    def read_bvh(self):
        i = self.info()
        keys = np.zeros(i.n_tris, dtype=np.uint64)
        prim = np.zeros(i.n_tris, dtype=np.uint32)
        nodes = np.zeros((i.n_nodes, 16), dtype=np.uint32)
        self.ctx._check(lib_amd().pt_scene_read_bvh(self.h, keys.ctypes.data, prim.ctypes.data, nodes.ctypes.data))
        return keys, prim, nodes

    def read_bvh4(self):
        nodes = np.zeros((self.info().n_wide_nodes, 32), dtype=np.uint32)
        self.ctx._check(lib_amd().pt_scene_read_bvh4(self.h, nodes.ctypes.data))
        return nodes

    def read_bvh8(self):
        """-> (nodes [n_wide8, 16] u32, prim_of_pos8 [n_tris] u32) of the 8-wide tree (include/pt_api.h: pt_scene_read_bvh8)"""
        self.ctx._check(lib_amd().pt_scene_read_bvh8(self.h, None, None))      # big scenes build their 8-wide nodes on first request
        i = self.info()
        nodes = np.zeros((i.n_wide8_nodes, 16), dtype=np.uint32)
        prim = np.zeros(i.n_tris, dtype=np.uint32)
        self.ctx._check(lib_amd().pt_scene_read_bvh8(self.h, nodes.ctypes.data, prim.ctypes.data))
        return nodes, prim

    def trace(self, rays6, tmin=0.001, tmax=10000.0, extend=EXTEND_AUTO):
        """Closest-hit query alone (traceRayEXT, raygen.rgen:63-75)."""
        r = np.ascontiguousarray(rays6, dtype=np.float32).reshape(-1, 6)
        hits = np.zeros(r.shape[0], dtype=HIT_DTYPE)
        self.ctx._check(lib_amd().pt_trace(self.h, r.ctypes.data, r.shape[0], tmin, tmax, extend, hits.ctypes.data))
        return hits

    def close(self):
        if self.h:
            lib_amd().pt_scene_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Film:
    """The storage image (main.cpp:481-484): float running mean + the reference's rgba8 image."""

    def __init__(self, ctx, width, height, device_ptr=None):
        self.ctx, self.width, self.height = ctx, width, height
        self.h = C.c_void_p()
        if device_ptr is None:
            rc = lib_amd().pt_film_create(ctx.h, width, height, C.byref(self.h))
        else:
            rc = lib_amd().pt_film_create_external(ctx.h, width, height, C.c_void_p(device_ptr), C.byref(self.h))
        ctx._check(rc)

    def clear(self):
        self.ctx._check(lib_amd().pt_film_clear(self.h))

    def read_f32(self):
        a = np.zeros((self.height, self.width, 3), dtype=np.float32)
        self.ctx._check(lib_amd().pt_film_read_f32(self.h, a.ctypes.data))
        return a

    def read_bgra8(self):
        a = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self.ctx._check(lib_amd().pt_film_read_bgra8(self.h, a.ctypes.data))
        return a

    def close(self):
        if self.h:
            lib_amd().pt_film_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """One rank's RCCL communicator for presenting tile-sharded renders (include/pt_api.h: pt_comm_*)."""

    def __init__(self, ctx, unique_id, world, rank):
        self.ctx, self.world, self.rank = ctx, world, rank
        self.h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        ctx._check(lib_amd().pt_comm_create(ctx.h, buf, world, rank, C.byref(self.h)))

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = lib_amd().pt_comm_unique_id(buf)
        if rc != PT_OK:
            raise PtError(rc, "pt_comm_unique_id (is RCCL installed?)")
        return buf.raw

    def ranks(self):
        n = C.c_uint32()
        self.ctx._check(lib_amd().pt_comm_ranks(self.h, C.byref(n)))
        return n.value

    def present(self, film, device_image_ptr, root=0):
        self.ctx._check(lib_amd().pt_film_present(film.h, self.h, root, C.c_void_p(device_image_ptr)))

    def close(self):
        if self.h:
            lib_amd().pt_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """hipMalloc'ed memory through the C-ABI (for callers without torch): .ptr, .read(dtype, shape), .write(array)"""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = C.c_void_p()
        ctx._check(lib_amd().pt_device_alloc(ctx.h, nbytes, C.byref(p)))
        self.ptr = p.value

    def read(self, dtype, shape):
        a = np.zeros(shape, dtype=dtype)
        assert a.nbytes <= self.nbytes
        self.ctx._check(lib_amd().pt_device_read(self.ctx.h, C.c_void_p(self.ptr), a.ctypes.data, a.nbytes))
        return a

    def write(self, array):
        a = np.ascontiguousarray(array)
        assert a.nbytes <= self.nbytes
        self.ctx._check(lib_amd().pt_device_write(self.ctx.h, C.c_void_p(self.ptr), a.ctypes.data, a.nbytes))

    def close(self):
        if self.ptr:
            lib_amd().pt_device_free(self.ctx.h, C.c_void_p(self.ptr))
            self.ptr = None


def film_tile_count(film, rank, world):
    n = C.c_uint32()
    film.ctx._check(lib_amd().pt_film_tile_count(film.h, rank, world, C.byref(n)))
    return n.value


def film_pack_tiles(film, rank, world, device_ptr):
    film.ctx._check(lib_amd().pt_film_pack_tiles(film.h, rank, world, C.c_void_p(device_ptr)))


def film_unpack_tiles(film, rank, world, device_packed_ptr, device_image_ptr):
    film.ctx._check(lib_amd().pt_film_unpack_tiles(film.h, rank, world, C.c_void_p(device_packed_ptr), C.c_void_p(device_image_ptr)))


def render(scene, film, params):
    """pushConstants(frame) + traceRaysKHR(W,H,1) + waitIdle (main.cpp:656-659, 683), for
    params.frame_count consecutive frames."""
    scene.ctx._check(lib_amd().pt_render(scene.h, film.h, C.byref(params)))


def render_prepare(scene, film, params):
    """Allocate the workspace pt_render would use for these params (no rendering)."""
    scene.ctx._check(lib_amd().pt_render_prepare(scene.h, film.h, C.byref(params)))
