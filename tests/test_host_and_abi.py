"""CPU tests of the host side (OBJ/MTL ingest, image output) and of the C-ABI surface."""
import os
import re

import numpy as np
import pytest

import obj_ref

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_loader_matches_independent_restatement(pt):
    v, i, f = pt.load_obj(pt.ASSET_CORNELL)
    rv, ri, rf = obj_ref.load_obj(pt.ASSET_CORNELL)
    assert v.dtype == np.float32 and i.dtype == np.uint32 and f.dtype == np.float32
    assert v.tobytes() == rv.tobytes() and i.tobytes() == ri.tobytes() and f.tobytes() == rf.tobytes()
    # main.cpp:42: Y negated; main.cpp:45: running indices
    assert (i == np.arange(108)).all()
    assert v.reshape(-1, 3)[:, 1].max() <= 0.0


def test_loader_ragged_inputs(pt, tmp_path):
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 0.1 0.2 0.3\nKe 1 2 3\nnewmtl grey\nKd 0.5\n")
    obj = tmp_path / "t.obj"
    obj.write_text(
        "# comment\nmtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 2 0 # trailing comment\n"
        "f 1 2 3\n"                 # no material yet -> default grey 0.6
        "usemtl a\nf 1/1/1 2/2/2 3/3/3 4/4/4 5/5/5\n"   # pentagon, v/vt/vn form -> 3 fan triangles
        "usemtl grey\nf -5//1 -4//1 -3//1\n"             # negative indices, v//vn form
        "usemtl nosuch\nf 1 2 3\n")                      # unknown material -> default
    v, i, f = pt.load_obj(str(obj))
    rv, ri, rf = obj_ref.load_obj(str(obj))
    assert v.tobytes() == rv.tobytes() and f.tobytes() == rf.tobytes()
    faces = f.reshape(-1, 6)
    assert faces.shape[0] == 6
    np.testing.assert_allclose(faces[0], [0.6, 0.6, 0.6, 0, 0, 0])
    np.testing.assert_allclose(faces[1], [0.1, 0.2, 0.3, 1, 2, 3])
    np.testing.assert_allclose(faces[4], [0.5, 0, 0, 0, 0, 0])   # `Kd 0.5`: missing components stay 0 (tinyobj parseReal3)
    np.testing.assert_allclose(faces[5], [0.6, 0.6, 0.6, 0, 0, 0])
    tri = v.reshape(-1, 3, 3)
    np.testing.assert_allclose(tri[3], [[0, 0, 0], [0, -1, 0], [0.5, -2, 0]])  # fan (0,3,4), y flipped


def test_loader_quad_rules(pt, tmp_path, orc):
    """fan (default, what every fixture uses) against the shorter-diagonal rule of newer tinyobjloader releases: a
    non-rectangular quad is cut along the other diagonal; a declared material without Kd is black (InitMaterial);
    on the Cornell box the two rules differ in 16 of 18 quads yet give the same image within the stated tolerance."""
    (tmp_path / "m.mtl").write_text("newmtl bare\nKe 1 1 1\n")
    obj = tmp_path / "q.obj"
    obj.write_text("mtllib m.mtl\nusemtl bare\nv 0 0 0\nv 1 0 0\nv 3 2 0\nv 0 1 0\nf 1 2 3 4\n")   # |v0v2|^2 = 13 > |v1v3|^2 = 2
    v, _, f = pt.load_obj(str(obj))
    tri = v.reshape(-1, 3, 3)
    np.testing.assert_allclose(tri[0], [[0, 0, 0], [1, 0, 0], [3, -2, 0]])
    np.testing.assert_allclose(f.reshape(-1, 6)[0], [0, 0, 0, 1, 1, 1])
    v2, _, _ = pt.load_obj(str(obj), flags=pt.QUAD_SHORTER_DIAGONAL)
    tri2 = v2.reshape(-1, 3, 3)
    np.testing.assert_allclose(tri2[0], [[0, 0, 0], [1, 0, 0], [0, -1, 0]])      # (0,1,3)
    np.testing.assert_allclose(tri2[1], [[1, 0, 0], [3, -2, 0], [0, -1, 0]])     # (1,2,3)
    a = pt.load_obj(pt.ASSET_CORNELL)
    b = pt.load_obj(pt.ASSET_CORNELL, flags=pt.QUAD_SHORTER_DIAGONAL)
    ta, tb = a[0].reshape(-1, 2, 3, 3), b[0].reshape(-1, 2, 3, 3)
    assert ta.shape == tb.shape and sum(not np.array_equal(x, y) for x, y in zip(ta, tb)) == 16
    p = orc.default_params(width=96, height=64, spp_per_frame=4, max_depth=8)
    ia, ra, _, _ = orc.Scene(*a).render_frame(p)
    ib, rb, _, _ = orc.Scene(*b).render_frame(p)
    d = np.linalg.norm(ia.astype(np.float64) - ib, axis=-1) / np.maximum(1.0, np.linalg.norm(ib.astype(np.float64), axis=-1))
    assert (d <= 1e-4).mean() >= 0.99 and abs(ra - rb) <= 0.002 * ra


def _random_obj_text(rng, n_lines, error=None):
    """OBJ text with everything the loader reads: comments, blank lines, CRLF ends, `v` in several number formats, faces of 3..6
    vertices with absolute / relative indices and the v/vt/vn forms, `usemtl` of known, unknown and later-overridden names,
    `mtllib` lines in the middle.  error: (kind, line) plants one bad line."""
    out, nv = [], 0
    names = ["a", "b", "c", "nosuch", "late"]
    fmt = [lambda x: "%.9g" % x, lambda x: "%+.9g" % x, lambda x: "%.8e" % x, lambda x: repr(float(x))]
    for no in range(1, n_lines + 1):
        r = rng.random()
        if error and no == error[1]:
            kind = error[0]
            line = {"vertex": "v 1 2", "vertex_text": "v 0.5 x 1", "face": "f 1 zz 3", "zero": "f 1 0 2", "range": f"f 1 2 {nv + 5}",
                    "range_neg": f"f -1 -2 -{nv + 3}", "short": "f 1 2", "range_before_bad": f"f {nv + 9} qq 1", "range_in_short": f"f {nv + 2} 1"}[kind]
            out.append(line)
            continue
        if nv < 3 or r < 0.45:
            x = np.float32(rng.uniform(-3, 3, 3))
            f = fmt[int(rng.integers(len(fmt)))]
            out.append("v " + " ".join(f(c) for c in x) + ("  # corner" if rng.random() < 0.1 else ""))
            nv += 1
        elif r < 0.8:
            k = int(rng.choice([3, 3, 3, 4, 4, 5, 6]))
            toks = []
            for _ in range(k):
                a = int(rng.integers(1, nv + 1))
                i = a if rng.random() < 0.5 else a - nv - 1
                form = int(rng.integers(4))
                toks.append([f"{i}", f"{i}/7", f"{i}//2", f"{i}/3/4"][form])
            out.append(("f " if rng.random() < 0.9 else "f\t") + " ".join(toks))
        elif r < 0.9:
            out.append("usemtl " + names[int(rng.integers(len(names)))])
        elif r < 0.93:
            out.append("mtllib " + str(rng.choice(["m1.mtl", "m2.mtl", "missing.mtl"])))
        elif r < 0.97:
            out.append(str(rng.choice(["", "# a comment", "g group", "vn 0 1 0", "s off", "   "])))
        else:
            out.append("usemtl")
    eol = "\r\n" if rng.random() < 0.3 else "\n"
    return eol.join(out) + (eol if rng.random() < 0.7 else "")


def test_loader_chunked_text_equals_one_reader(pt, tmp_path):
    """The loader reads big files with one thread per chunk of text; PTH_SMALL_CHUNKS makes that happen on short ones (chunks of
    ~256 bytes, so relative indices, materials and the quad rule reach across chunk boundaries).  Arrays -- and for a broken file
    the message and its line number -- equal a line-by-line reader's (tests/obj_ref.py load_obj_strict), with and without chunks."""
    (tmp_path / "m1.mtl").write_text("newmtl a\nKd 0.1 0.2 0.3\nKe 1 2 3\nnewmtl b\nKd 0.5\n")
    (tmp_path / "m2.mtl").write_text("newmtl late\nKd 0.9 0.8 0.7\nnewmtl a\nKd 0.25 0.5 0.75\nKe 4 5 6\n")
    rng = np.random.default_rng(12)
    kinds = [None, None, "vertex", "vertex_text", "face", "zero", "range", "range_neg", "short", "range_before_bad", "range_in_short"]
    for case in range(60):
        n_lines = int(rng.integers(5, 400))
        kind = kinds[case % len(kinds)]
        err = (kind, int(rng.integers(4, n_lines + 1))) if kind else None
        obj = tmp_path / f"r{case}.obj"
        with open(obj, "w", newline="") as f:
            f.write(_random_obj_text(rng, n_lines, err))
        for quad in (False, True):
            try:
                want = obj_ref.load_obj_strict(str(obj), quad)
            except obj_ref.ObjError as e:
                want = str(e)
            for flags in (0, pt.SMALL_CHUNKS):
                try:
                    got = pt.load_obj(str(obj), flags=flags | (pt.QUAD_SHORTER_DIAGONAL if quad else 0))
                except RuntimeError as e:
                    got = str(e).split(": ", 1)[1]
                if isinstance(want, str):
                    assert got == want, (case, kind, flags)
                else:
                    assert not isinstance(got, str), (case, got)
                    assert all(x.tobytes() == y.tobytes() for x, y in zip(got, want)), (case, flags, quad)


def test_make_soup_equals_the_obj_round_trip(pt, tmp_path):
    path = str(tmp_path / "s.obj")
    pt.write_soup_obj(path, 3000, 7)
    a, b = pt.load_obj(path), pt.make_soup(3000, 7)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b))


def test_loader_errors(pt, tmp_path):
    with pytest.raises(RuntimeError):
        pt.load_obj(str(tmp_path / "missing.obj"))
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nf 1 2 9\n")
    with pytest.raises(RuntimeError, match="out of range"):
        pt.load_obj(str(bad))
    # a face token with trailing garbage is a bad face, as for the one-reader restatement (strtol alone would read "1x" as index 1: ADVICE r04)
    import obj_ref
    for tok in ("1x", "2/3x/4", "3.5", "0x2", "1/"):
        junk = tmp_path / "junk.obj"
        junk.write_text(f"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 {tok}\n")
        ok_ref = True
        try:
            obj_ref.load_obj_strict(str(junk))
        except obj_ref.ObjError:
            ok_ref = False
        if ok_ref:           # ("1/", "2/3x/4": the vertex index before the first '/' is all that either reader looks at)
            pt.load_obj(str(junk))
        else:
            with pytest.raises(RuntimeError, match="bad face"):
                pt.load_obj(str(junk))
    empty = tmp_path / "empty.obj"
    empty.write_text("v 0 0 0\n")
    with pytest.raises(RuntimeError, match="no faces"):
        pt.load_obj(str(empty))
    # missing MTL is only a warning in tinyobjloader: faces get the default material
    nomtl = tmp_path / "nomtl.obj"
    nomtl.write_text("mtllib nothere.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl x\nf 1 2 3\n")
    _, _, f = pt.load_obj(str(nomtl))
    np.testing.assert_allclose(f, [0.6, 0.6, 0.6, 0, 0, 0])
    # `mtllib` / `usemtl` without a name: the "library" is then the OBJ's directory, which opens for reading and reports LONG_MAX bytes -- found by
    # tests/test_sanitizers.py as a std::bad_alloc that left the C-ABI; it is a missing library like any other
    noname = tmp_path / "noname.obj"
    noname.write_text("mtllib\nusemtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    _, _, f = pt.load_obj(str(noname))
    np.testing.assert_allclose(f, [0.6, 0.6, 0.6, 0, 0, 0])


def test_soup_generator_roundtrip(pt, tmp_path):
    path = str(tmp_path / "soup.obj")
    pt.write_soup_obj(path, 2000, 1)
    v, i, f = pt.load_obj(path)
    rv, ri, rf = obj_ref.load_obj(path)
    assert v.tobytes() == rv.tobytes() and f.tobytes() == rf.tobytes()
    assert i.size == 6000
    tri = v.reshape(-1, 3, 3)
    assert tri[..., 0].min() > -1.02 and tri[..., 0].max() < 1.02
    assert tri[..., 1].min() > -2.02 and tri[..., 1].max() < 0.02     # Y negated at load
    faces = f.reshape(-1, 6)
    emit = faces[:, 3:].sum(1) > 0
    assert (np.nonzero(emit)[0] % 64 == 7).all() and emit.sum() == len(range(7, 2000, 64))
    pt.write_soup_obj(str(tmp_path / "soup2.obj"), 2000, 1)          # deterministic
    body = lambda fn: [l for l in open(fn) if not l.startswith(("#", "mtllib"))]
    assert body(path) == body(str(tmp_path / "soup2.obj"))


def test_image_writers(pt, tmp_path):
    bgra = np.zeros((2, 3, 4), np.uint8)
    bgra[..., 0], bgra[..., 1], bgra[..., 2], bgra[..., 3] = 10, 20, 30, 255
    pt.write_ppm(str(tmp_path / "a.ppm"), bgra)
    raw = open(tmp_path / "a.ppm", "rb").read()
    assert raw.startswith(b"P6\n3 2\n255\n") and raw[-3:] == bytes([30, 20, 10])
    rgb = np.arange(18, dtype=np.float32).reshape(2, 3, 3)
    pt.write_pfm(str(tmp_path / "a.pfm"), rgb)
    raw = open(tmp_path / "a.pfm", "rb").read()
    body = np.frombuffer(raw[len(b"PF\n3 2\n-1.0\n"):], np.float32).reshape(2, 3, 3)
    assert (body[::-1] == rgb).all()


def _declared(header, prefix):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s_\w+)\s*\(" % prefix, text)))


def test_abi_exports_every_declared_symbol(pt):
    api = _declared("pt_api.h", "pt")
    host = _declared("pt_host.h", "pth")
    assert sorted(pt.API_SYMBOLS) == api
    assert sorted(pt.HOST_SYMBOLS) == host
    La, Lh = pt.lib_amd(), pt.lib_host()
    for s in api:
        assert hasattr(La, s), s
    for s in host:
        assert hasattr(Lh, s), s


def test_struct_layouts_match_header(pt, tmp_path):
    """The ctypes mirrors against include/pt_api.h itself: gcc prints sizeof / offsetof of the header's structs."""
    import ctypes as C
    import shutil
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pt_api.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %d\\n",'
                   'sizeof(pt_tuning), sizeof(pt_params), sizeof(pt_stats), sizeof(pt_scene_info), offsetof(pt_params, sample_groups),'
                   'offsetof(pt_stats, workspace_bytes), offsetof(pt_stats, wave_refills), offsetof(pt_scene_info, tree_area_ploc), offsetof(pt_stats, pipeline), PT_API_VERSION);return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call([shutil.which("gcc") or "gcc", "-I", os.path.join(REPO, "include"), "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    assert got == [C.sizeof(pt.Tuning), C.sizeof(pt.Params), C.sizeof(pt.Stats), C.sizeof(pt.SceneInfo), pt.Params.sample_groups.offset,
                   pt.Stats.workspace_bytes.offset, pt.Stats.wave_refills.offset, pt.SceneInfo.tree_area_ploc.offset, pt.Stats.pipeline.offset, 6], got
    assert C.sizeof(pt.Tuning) == 4 * 32 and pt.PIPELINE_FUSED == 2 and pt.PIPELINE_AUTO == 3
    assert C.sizeof(pt.Params) == 4 * 8 + 4 * 9 + 4 * 7
    p = pt.library_default_params()   # (pt_params_default itself; the suite's own pt.default_params names the wavefront pipeline, conftest.py)
    assert (p.width, p.height, p.spp_per_frame, p.max_depth, p.world, p.frame_count) == (1024, 1024, 32, 8, 1, 1)
    assert list(p.cam_origin) == [0.0, -1.0, 5.0] and list(p.cam_target) == [0.0, -1.0, 2.0]
    assert [round(x, 6) for x in p.env] == [0.7, 0.6, 0.5]
    assert abs(p.tmin - 0.001) < 1e-9 and p.tmax == 10000.0
    assert p.pipeline == pt.PIPELINE_AUTO   # what a caller who follows INTEGRATION.md gets: the fastest bit-exact pipeline for the scene


def test_no_cpu_fallback(pt):
    """Without a GPU the product must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pt.PtError) as e:
        pt.Context(0)
    assert e.value.status == 2  # PT_ERR_NO_DEVICE


def test_abi_is_null_safe_without_a_gpu(pt):
    """Every entry point must reject null handles with PT_ERR_INVALID_ARG (or be a no-op for the
    destroy functions) -- callable without a GPU, nothing may crash."""
    import ctypes as C
    L = pt.lib_amd()
    L.pt_ctx_destroy(None)
    L.pt_scene_destroy(None)
    L.pt_film_destroy(None)
    L.pt_params_default(None)
    assert L.pt_last_error(None) is not None
    null = C.c_void_p()
    out = C.c_void_p()
    assert L.pt_sync(None) == 1
    assert L.pt_scene_create(None, None, 0, None, 0, None, C.byref(out)) == 1
    assert L.pt_scene_set_instances(None, None, 0) == 1
    assert L.pt_scene_set_bvh_quality(None, 0) == 1
    assert L.pt_scene_get_info(None, None) == 1
    assert L.pt_scene_read_bvh(None, None, None, None) == 1
    assert L.pt_scene_read_bvh4(None, None) == 1
    assert L.pt_scene_read_bvh8(None, None, None) == 1
    assert L.pt_film_create(None, 4, 4, C.byref(out)) == 1
    assert L.pt_film_create_external(None, 4, 4, None, C.byref(out)) == 1
    assert L.pt_film_clear(None) == 1 and L.pt_film_read_f32(None, None) == 1 and L.pt_film_read_bgra8(None, None) == 1
    p = pt.default_params()
    assert L.pt_render(None, None, C.byref(p)) == 1 and L.pt_render_prepare(None, None, C.byref(p)) == 1
    assert L.pt_trace(None, None, 0, C.c_float(0), C.c_float(1), 0, None) == 1
    assert L.pt_get_stats(None, None) == 1 and L.pt_reset_stats(None) == 1
    assert L.pt_ctx_create(0, None, None) == 1
    assert L.pt_comm_unique_id(None) == 1 and L.pt_comm_create(None, None, 1, 0, C.byref(out)) == 1
    assert L.pt_comm_ranks(None, None) == 1 and L.pt_film_present(None, None, 0, None) == 1
    L.pt_comm_destroy(None)
    n = C.c_uint32()
    assert L.pt_film_tile_count(None, 0, 1, C.byref(n)) == 1 and L.pt_film_pack_tiles(None, 0, 1, None) == 1
    assert L.pt_film_unpack_tiles(None, 0, 1, None, None) == 1
    assert L.pt_device_alloc(None, 16, C.byref(out)) == 1 and L.pt_device_free(None, None) == 1
    assert L.pt_device_read(None, None, None, 0) == 1
    assert out.value is None and null.value is None
    H = pt.lib_host()
    err = C.create_string_buffer(64)
    assert H.pth_load_obj(None, None, None, err, 64) != 0
    H.pth_free_scene(None)
    assert H.pth_write_ppm_bgra8(None, None, 0, 0) != 0 and H.pth_write_pfm(None, None, 0, 0) != 0
    assert H.pth_write_soup_obj(None, 0, 0) != 0


def test_bench_gpus_n_needs_no_launcher():
    """`python bench.py --gpus 2` with no launcher environment must not die for launcher reasons.  It
    starts its own ranks (one process per GPU, rendezvous on 127.0.0.1); without a GPU each rank says so and the parent
    stops the rest and returns their status -- within seconds, no hang.  (On a GPU box: rank 1 lacks a device, or both run.)"""
    import subprocess
    import sys
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PT_BENCH_EMULATE")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--reps", "1",
                        "--width", "64", "--height", "64", "--no-cpu-baseline", "--no-extra-legs", "--full-line"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert "must be launched with" not in r.stderr
    assert time.time() - t0 < 240
    if r.returncode != 0:
        assert ("needs a GPU" in r.stderr or "needs GPU 1" in r.stderr) and "stopping the other ranks" in r.stderr, r.stderr[-1500:]
    else:
        assert '"n_gpus": 2' in r.stdout


def test_product_never_touches_the_oracle():
    pkg = os.path.join(REPO, "single-file-vulkan-pathtracing_amd")
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(root, fn), errors="ignore").read()
                assert "pt_oracle" not in text and "oracle/" not in text, os.path.join(root, fn)


def test_fast_division_by_the_pdf_is_proven_for_every_float(tmp_path):
    """csrc/pt_math.h div3_by_pdf replaces three true divisions by 1/(2 pi) with mul + 2 fma inside a guarded
    range; tests/exhaustive_div_by_pdf.c tries EVERY float of that range against the IEEE quotient."""
    import shutil
    import subprocess
    if "fma" not in open("/proc/cpuinfo").read().split():
        pytest.skip("host CPU without FMA3: fmaf would be emulated, the enumeration takes too long")
    exe = tmp_path / "exh"
    subprocess.check_call([shutil.which("gcc") or "gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", str(exe),
                           os.path.join(os.path.dirname(os.path.abspath(__file__)), "exhaustive_div_by_pdf.c"), "-lpthread", "-lm"])
    out = subprocess.run([str(exe), str(min(os.cpu_count() or 1, 16))], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches 0" in out.stdout and "tried 3690987522" in out.stdout


def test_bench_counter_csv_sums_and_the_fallback(tmp_path, monkeypatch):
    """bench.py measures `roofline.traffic` itself: the same frames under `rocprofv3 --pmc` in a child process.  What can be checked without
    a GPU: the sum over rocprofv3's counter CSVs (kernels by short-name prefix, the instrumented instantiations left out, dispatches
    counted once), and that a run which is itself being profiled, or whose first pass failed, does not start (more) passes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    d = tmp_path / "pass" / "host" / "1234"
    d.mkdir(parents=True)
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Kernel_Name,Counter_Name,Counter_Value",
            '1,1,0,"(anonymous namespace)::k_extend_lds7p(HIP_vector_type<float, 4u> const*, int)",FETCH_SIZE,100',
            '2,2,0,"(anonymous namespace)::k_extend_lds7p(HIP_vector_type<float, 4u> const*, int)",FETCH_SIZE,140',
            '2,2,0,"(anonymous namespace)::k_extend_lds7p(HIP_vector_type<float, 4u> const*, int)",FETCH_SIZE,60',     # a second row of the same dispatch (another XCD)
            '3,3,0,"void (anonymous namespace)::k_extend<true, true, false, true, false>(HIP_vector_type<float, 4u> const*)",FETCH_SIZE,999',   # the counting instantiation
            '4,4,0,"void (anonymous namespace)::k_shade<2, true, false, false>(ptw::RenderConst, unsigned int const*)",FETCH_SIZE,50',
            '5,5,0,"void (anonymous namespace)::k_extend8<false, false, 7>(HIP_vector_type<unsigned int, 4u> const*)",WRITE_SIZE,7',
            '6,6,0,"(anonymous namespace)::k_resolve(ptw::RenderConst)",FETCH_SIZE,5']
    (d / "p_counter_collection.csv").write_text("\n".join(rows) + "\n")
    tot, n = b.sum_counter_csvs(str(tmp_path / "pass"), "FETCH_SIZE", ("k_extend", "k_shade"))
    assert tot == {"k_extend": 300.0, "k_shade": 50.0} and n == {"k_extend": 2, "k_shade": 1}
    tot, n = b.sum_counter_csvs(str(tmp_path / "pass"), "WRITE_SIZE", ("k_extend",))
    assert tot == {"k_extend": 7.0} and n == {"k_extend": 1}
    # no profiler inside a profiler; and no second try after a failure
    monkeypatch.setenv("ROCPROFILER_LIBRARY_CTOR", "1")
    assert b.live_traffic(["--pmc-child"]) is None and b._LIVE_PMC_FAILED[0] is False
    monkeypatch.delenv("ROCPROFILER_LIBRARY_CTOR")
    b._LIVE_PMC_FAILED[0] = True
    assert b.live_traffic(["--pmc-child"]) is None
    r = {}
    b.apply_live_traffic(r, None, None, 1.0, 1)
    assert "not measured" in r["traffic_live"]["source"]
