#!/usr/bin/env python3
"""Known answers at BASELINE's FULL sizes: the CPU oracle renders frame 0 of configs C2, C4 and C5 exactly as BASELINE.json
states them (1920x1080; C2/C4 32 spp depth 8, C5 16 spp depth 16) and this script records the exact ray count and the SHA-256
of the float32 film (H x W x 3, row-major) in tests/golden/fullsize_hashes.json.  `tests/test_gpu_parity.py::
test_full_size_frames_equal_the_oracle_known_answers` renders the same frames on the GPU and compares both -- a bit-exact
check of 6.2 M floats per config that needs no oracle run on the GPU box.  (bench.py compares the same frames against a
LIVE oracle run inside its cpu_baseline leg; the numbers agree: C2 224 112 445 rays, C4 163 213 621, C5 118 193 858.)

    python tests/golden/make_fullsize_hashes.py [c2 c4 c5]      # ~2 min for C2+C4, ~10 min for C5 on 8 cores
    python tests/golden/make_fullsize_hashes.py c3              # ~20 min on 8 cores

    python tests/golden/make_fullsize_hashes.py ref1024         # ~2 min on 8 cores

`ref1024` is the REFERENCE'S OWN launch (main.cpp:16-17 WIDTH = HEIGHT = 1024, main.cpp:659 traceRaysKHR(WIDTH, HEIGHT, 1), 32 spp, depth 8: exactly
what pt_params_default returns), frames 0..3 blended progressively as its frame loop does (main.cpp:647-685, raygen.rgen:88-90): the exact ray count of
every frame and the SHA-256 of the float film and of the bgra8 storage image after each of them; `test_reference_dispatch_1024_*` issues the same four
blocking calls on the GPU through PT_PIPELINE_AUTO and requires all of it to the bit.

`c3` is BASELINE config C3 at its size: the Cornell box, 1920x1080, 1024 spp = frames 0..31 of 32 spp (seed multipliers
m = 1..1024, raygen.rgen:47), depth 8, blended progressively as raygen.rgen:88-90 does -- the float film AND the reference's rgba8
storage image.  Recorded: the exact ray count of every frame (their sum is the 8-GPU job's total), the SHA-256 of the float
film and of the bgra8 image after frames 0, 1, 3, 7, 15 and 31, and for the world-8 decomposition (8x8 tiles, tile (tx,ty) ->
rank (tx+ty) % 8) the exact number of rays each rank traces: `test_c3_*` renders the same 32 frames on
the GPU as one device and as the eight ranks one after the other and requires all of it to the bit.
"""
import hashlib
import importlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import pt_oracle as orc  # noqa: E402

pt = importlib.import_module("single-file-vulkan-pathtracing_amd")   # host loader + the frozen C4 / C5 recipes only (no GPU)
OUT = os.path.join(HERE, "fullsize_hashes.json")
CONFIGS = {
    "c2": dict(spp=32, depth=8, scene="cornell"),
    "c4": dict(spp=32, depth=8, scene="cornell", instances=True),
    "c5": dict(spp=16, depth=16, scene="soup", n_tris=1000000, seed=1),
}


C3_FRAMES = 32
C3_MARKS = (0, 1, 3, 7, 15, 31)
C3_WORLD = 8


def make_c3(res):
    import numpy as np
    osc = orc.Scene(*pt.load_obj(os.path.join(REPO, "assets", "CornellBox-Original.obj")))
    w, h = 1920, 1080
    film = np.zeros((h, w, 3), np.float32)
    bgra = np.zeros((h, w, 4), np.uint8)
    # rank of every pixel under the world-8 tile decomposition (csrc/film_work.hip ptw_ensure_work: (tx + ty) % world)
    ty, tx = np.meshgrid(np.arange(h) // 8, np.arange(w) // 8, indexing="ij")
    rank_of = ((tx + ty) % C3_WORLD).astype(np.int64)
    rays_frame, rays_rank = [], np.zeros(C3_WORLD, np.int64)
    marks = {}
    t0 = time.perf_counter()
    for frame in range(C3_FRAMES):
        p = orc.default_params(width=w, height=h, spp_per_frame=32, max_depth=8, frame=frame)
        ray_map = np.zeros((h, w), np.uint32)
        img, rays, _, _ = osc.render_frame(p, mode=1, nthreads=os.cpu_count() or 1, ray_map=ray_map)
        assert int(ray_map.sum(dtype=np.int64)) == rays
        rays_rank += np.bincount(rank_of.ravel(), weights=ray_map.ravel().astype(np.float64), minlength=C3_WORLD).astype(np.int64)
        orc.accumulate_f32(film, img, frame)
        orc.accumulate_bgra8(bgra, img, frame)
        rays_frame.append(int(rays))
        if frame in C3_MARKS:
            marks[str(frame)] = {"film_sha256": hashlib.sha256(film.astype("<f4").tobytes()).hexdigest(),
                                 "bgra8_sha256": hashlib.sha256(bgra.tobytes()).hexdigest(),
                                 "rays_so_far": int(sum(rays_frame))}
        print("c3 frame", frame, rays, round(time.perf_counter() - t0, 1), "s", flush=True)
    res["c3"] = {"width": w, "height": h, "spp_per_frame": 32, "max_depth": 8, "frames": C3_FRAMES, "world": C3_WORLD,
                 "rays": int(sum(rays_frame)), "rays_per_frame": rays_frame, "after_frame": marks,
                 "rays_per_rank_world8": [int(x) for x in rays_rank],
                 "film_sum_f64": float(film.astype("float64").sum()), "scene": "CornellBox-Original.obj",
                 "oracle_seconds": round(time.perf_counter() - t0, 1)}
    print("c3", {k: v for k, v in res["c3"].items() if k != "rays_per_frame"}, flush=True)


def make_ref1024(res):
    import numpy as np
    osc = orc.Scene(*pt.load_obj(os.path.join(REPO, "assets", "CornellBox-Original.obj")))
    p0 = orc.default_params()
    w, h = int(p0.width), int(p0.height)
    assert (w, h, int(p0.spp_per_frame), int(p0.max_depth)) == (1024, 1024, 32, 8), "the oracle's defaults are the reference's constants"
    film = np.zeros((h, w, 3), np.float32)
    bgra = np.zeros((h, w, 4), np.uint8)
    frames = []
    t0 = time.perf_counter()
    for frame in range(4):
        img, rays, _, _ = osc.render_frame(orc.default_params(frame=frame), mode=1, nthreads=os.cpu_count() or 1)
        orc.accumulate_f32(film, img, frame)
        orc.accumulate_bgra8(bgra, img, frame)
        frames.append({"frame": frame, "rays": int(rays), "film_sha256": hashlib.sha256(film.astype("<f4").tobytes()).hexdigest(),
                       "bgra8_sha256": hashlib.sha256(bgra.tobytes()).hexdigest()})
        print("ref1024 frame", frame, rays, round(time.perf_counter() - t0, 1), "s", flush=True)
    res["ref1024"] = {"width": w, "height": h, "spp_per_frame": 32, "max_depth": 8, "frames": frames, "rays": int(sum(f["rays"] for f in frames)),
                      "film_sum_f64": float(film.astype("float64").sum()), "scene": "CornellBox-Original.obj",
                      "what": "the reference's own dispatch (main.cpp:16-17, 659), one blocking launch per frame (main.cpp:647-685)",
                      "oracle_seconds": round(time.perf_counter() - t0, 1)}


def main():
    want = sys.argv[1:] or list(CONFIGS)
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    if "ref1024" in want:
        make_ref1024(res)
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
        want = [x for x in want if x != "ref1024"]
    if "c3" in want:
        make_c3(res)
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
        want = [x for x in want if x != "c3"]
    for name in want:
        c = CONFIGS[name]
        arrays = (pt.load_obj(os.path.join(REPO, "assets", "CornellBox-Original.obj")) if c["scene"] == "cornell"
                  else pt.make_soup(c["n_tris"], c["seed"]))
        osc = orc.Scene(*arrays)
        if c.get("instances"):
            osc.set_instances(pt.cornell_grid_instances())
        p = orc.default_params(width=1920, height=1080, spp_per_frame=c["spp"], max_depth=c["depth"])
        t0 = time.perf_counter()
        img, rays, _, _ = osc.render_frame(p, mode=1, nthreads=os.cpu_count() or 1)
        dt = time.perf_counter() - t0
        res[name] = {"width": 1920, "height": 1080, "spp_per_frame": c["spp"], "max_depth": c["depth"], "frame": 0,
                     "rays": int(rays), "film_sha256": hashlib.sha256(img.astype("<f4").tobytes()).hexdigest(),
                     "film_sum_f64": float(img.astype("float64").sum()),
                     "scene": ("CornellBox-Original.obj" + (" x cornell_grid_instances()" if c.get("instances") else "")) if c["scene"] == "cornell"
                              else f"pth_make_soup({c['n_tris']}, seed {c['seed']})",
                     "oracle_seconds": round(dt, 1)}
        print(name, res[name], flush=True)
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
