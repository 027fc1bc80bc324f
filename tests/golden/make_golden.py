"""Regenerates tests/golden/c1_256_1spp_d4.npz from the CPU oracle (run from the repo root).

The reference itself cannot run here (needs a Vulkan RT driver), so this golden only guards the
ORACLE against drift: config C1 of BASELINE.json (CornellBox-Original.obj, 256x256, 1 spp, depth 4).
The scene arrays come from tests/obj_ref.py (independent Python restatement of main.cpp:28-58).
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import obj_ref  # noqa: E402
from oracle import pt_oracle as O  # noqa: E402

v, i, f = obj_ref.load_obj(os.path.join(REPO, "assets", "CornellBox-Original.obj"))
sc = O.Scene(v, i, f)
p = O.default_params(width=256, height=256, spp_per_frame=1, max_depth=4)
img, rays, cnt, fh = sc.render_frame(p, mode=0, nthreads=1, want_first_hits=True)
prim = fh["prim"].astype(np.int64)
prim[prim == O.MISS] = 255
np.savez_compressed(os.path.join(REPO, "tests", "golden", "c1_256_1spp_d4.npz"), image=img,
                    first_prim=prim.astype(np.uint8).reshape(256, 256), first_u=fh["u"].reshape(256, 256),
                    first_v=fh["v"].reshape(256, 256), rays=np.int64(rays))
print("rays", rays, "mean", img.reshape(-1, 3).mean(0))
