"""The proof behind pt_tuning.cull, checked on the CPU with the oracle (no GPU): no pixel outside the projection of the scene's box -- a pixel of
slack on every side -- has a camera ray that hits anything.  `subject_rect` below restates csrc/render.hip subject_rect (same operations in
float32); the oracle's per-pixel ray map says which pixels had a primary hit (a hit spawns a bounce ray, so such a pixel traces more than spp
rays at max_depth >= 2).  The GPU suite checks the library's own rectangle through its effect: films and ray counts with and without the cull
are the oracle's (tests/test_gpu_parity.py test_fused_subject_first_order_changes_no_bit, test_fused_cull_on_two_level_scenes)."""
import numpy as np
import pytest

f32 = np.float32


def subject_rect(bmin, bmax, cam_origin, cam_target, width, height):
    """-> (x0, y0, x1, y1) in pixels, or None where the library culls nothing (a corner of the box is not in front of the origin, or the box is
    off the image)."""
    o, t = np.asarray(cam_origin, f32), np.asarray(cam_target, f32)
    den = f32(t[2] - o[2])
    if not abs(den) > 0:
        return None
    lo, hi = [f32(3.0e38)] * 2, [f32(-3.0e38)] * 2
    for c in range(8):
        P = [f32(bmax[k] if (c >> k) & 1 else bmin[k]) for k in range(3)]
        a = f32(f32(P[2] - o[2]) / den)
        if not a > f32(1.0e-4):
            return None
        for k in range(2):
            d = f32(f32(o[k] + f32(f32(P[k] - o[k]) / a)) - t[k])
            lo[k], hi[k] = min(lo[k], d), max(hi[k], d)
    r = [0, 0, 0, 0]
    for k, size in ((0, f32(width)), (1, f32(height))):
        a = f32(f32(f32(lo[k] + f32(1)) * f32(0.5)) * size) - f32(1)
        b = f32(f32(f32(hi[k] + f32(1)) * f32(0.5)) * size) + f32(1)
        if not (a == a and b == b) or b < 0 or a > size:
            return None
        r[k] = int(max(a, f32(0)))
        r[k + 2] = int(min(b, size - f32(1)))
    return tuple(r)


def _views(rng, n):
    yield dict()                                                                    # the reference's camera
    yield dict(cam_origin=(1.1, -1.0, 5.0), cam_target=(1.1, -1.0, 2.0))            # the box at the left
    yield dict(cam_origin=(1.0, -0.2, 5.0), cam_target=(1.0, -0.2, 2.0))            # in a corner
    yield dict(cam_origin=(0.0, -1.0, 9.0), cam_target=(0.0, -1.0, 6.0))            # far
    yield dict(cam_origin=(0.0, -1.0, 2.2), cam_target=(0.0, -1.0, -0.8))           # close: the box larger than the image
    yield dict(cam_origin=(0.0, -1.0, 5.0), cam_target=(0.0, -1.0, 4.7))            # a short view axis: the image plane near the origin
    for _ in range(n):
        o = rng.uniform(-2.5, 2.5, 3).astype(f32) + f32([0, -1, 0])
        o[2] = f32(rng.uniform(1.5, 9.0))
        t = o + rng.uniform(-0.6, 0.6, 3).astype(f32)
        t[2] = o[2] - f32(rng.uniform(0.5, 4.0))
        yield dict(cam_origin=tuple(float(x) for x in o), cam_target=tuple(float(x) for x in t))


def _scenes(pt, cornell_arrays):
    yield "cornell", cornell_arrays
    rng = np.random.default_rng(11)
    n = 300
    c = rng.uniform(-0.7, 0.7, (n, 1, 3)).astype(f32) + f32([0.2, -1.1, 0.1])
    v = (c + rng.uniform(-0.08, 0.08, (n, 3, 3)).astype(f32)).reshape(-1)
    i = np.arange(3 * n, dtype=np.uint32)
    f = np.tile(f32([0.7, 0.7, 0.7, 0, 0, 0]), n)
    yield "soup", (v, i, f)


def test_no_pixel_outside_the_rectangle_has_a_primary_hit(pt, orc, cornell_arrays):
    w, h, spp = 96, 54, 3
    rng = np.random.default_rng(5)
    checked = culled_pixels = 0
    for name, arrays in _scenes(pt, cornell_arrays):
        osc = orc.Scene(*arrays)
        used = np.asarray(arrays[0], f32).reshape(-1, 3)[np.unique(np.asarray(arrays[1]))]
        bmin, bmax = used.min(0), used.max(0)
        for cam in _views(rng, 14):
            p = orc.default_params(width=w, height=h, spp_per_frame=spp, max_depth=2, **cam)
            rect = subject_rect(bmin, bmax, p.cam_origin, p.cam_target, w, h)
            if rect is None:
                continue
            hit = np.zeros((h, w), bool)
            for frame in range(3):                      # (other jitters of the same pixels)
                p.frame = frame
                rays = np.zeros((h, w), np.uint32)
                osc.render_frame(p, ray_map=rays)
                hit |= rays > spp                       # a camera ray that hits spawns a bounce ray
            x0, y0, x1, y1 = rect
            outside = np.ones((h, w), bool)
            outside[max(y0, 0):y1 + 1, max(x0, 0):x1 + 1] = False
            assert not (hit & outside).any(), (name, cam, rect, np.argwhere(hit & outside)[:4])
            checked += 1
            culled_pixels += int(outside.sum())
            # ... and the rectangle is not idle: it hugs the hits to within the slack and a pixel's jitter wherever the box's outline is
            # its own geometry (the Cornell box's walls)
            if name == "cornell" and hit.any():
                ys, xs = np.nonzero(hit)
                assert xs.min() - x0 <= 3 and x1 - xs.max() <= 3 and ys.min() - y0 <= 3 and y1 - ys.max() <= 3, (cam, rect, xs.min(), xs.max(), ys.min(), ys.max())
    assert checked >= 20 and culled_pixels > 0
