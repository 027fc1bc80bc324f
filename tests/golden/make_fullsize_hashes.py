#!/usr/bin/env python3
"""Known answers at BASELINE's FULL sizes: the CPU oracle renders frame 0 of configs C2, C4 and C5 exactly as BASELINE.json
states them (1920x1080; C2/C4 32 spp depth 8, C5 16 spp depth 16) and this script records the exact ray count and the SHA-256
of the float32 film (H x W x 3, row-major) in tests/golden/fullsize_hashes.json.  `tests/test_gpu_parity.py::
test_full_size_frames_equal_the_oracle_known_answers` renders the same frames on the GPU and compares both -- a bit-exact
check of 6.2 M floats per config that needs no oracle run on the GPU box.  (bench.py compares the same frames against a
LIVE oracle run inside its cpu_baseline leg; the numbers agree: C2 224 112 445 rays, C4 163 213 621, C5 118 193 858.)

    python tests/golden/make_fullsize_hashes.py [c2 c4 c5]      # ~2 min for C2+C4, ~10 min for C5 on 8 cores
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


def main():
    want = sys.argv[1:] or list(CONFIGS)
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
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
