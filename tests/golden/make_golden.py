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

# C2 crop: BASELINE config 2 (1920x1080, 32 spp per frame, depth 8), frames 0 and 1, the 96x64-pixel
# rectangle at (912, 508) that straddles the tall box, the short box and the back wall.
C2_RECT = (912, 508, 96, 64)
crops, rays2 = [], []
for frame in (0, 1):
    p2 = O.default_params(width=1920, height=1080, spp_per_frame=32, max_depth=8, frame=frame)
    img2, r2 = O.render_rect(sc, p2, *C2_RECT)
    crops.append(img2)
    rays2.append(r2)
np.savez_compressed(os.path.join(REPO, "tests", "golden", "c2_crop_1080p_32spp_d8.npz"), rect=np.array(C2_RECT),
                    frame0=crops[0], frame1=crops[1], rays=np.array(rays2, dtype=np.int64))
print("c2 crop rays", rays2, "mean", crops[0].reshape(-1, 3).mean(0))
