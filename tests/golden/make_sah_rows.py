#!/usr/bin/env python3
"""Regression fixture of the device surface-area BVH4 builder (csrc/bvh4_sah_device.hip): the 128-B rows of the tree it builds
for the Cornell box and for a 700-triangle soup, as read back through pt_scene_read_bvh4 ON A GPU BOX.  The host builder that
used to cross-check it bit for bit was removed in round 3; the soundness tests (`test_sah_tree_*`) check that any
tree is a correct BVH, this fixture that the builder still makes THE tree its tuning was measured on.
    python tests/golden/make_sah_rows.py        # writes tests/golden/sah_rows.npz"""
import importlib, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")


def soup(n, seed, spread=0.1):   # tests/test_gpu_parity.py _soup
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
    v = (c + rng.uniform(-spread, spread, (n, 3, 3)).astype(np.float32)).astype(np.float32)
    faces = rng.uniform(0, 1, (n, 6)).astype(np.float32)
    faces[:, 3:] *= (rng.uniform(0, 1, (n, 1)) < 0.1)
    return v.reshape(-1), np.arange(3 * n, dtype=np.uint32), faces.reshape(-1).astype(np.float32)


if __name__ == "__main__":
    ctx = pt.Context(0)
    out = {}
    for name, arrays in (("cornell", pt.load_obj(pt.ASSET_CORNELL)), ("soup700", soup(700, 11))):
        sc = pt.Scene(ctx, *arrays)
        assert sc.info().bvh4_builder == 1
        out[name] = sc.read_bvh4()
        sc.close()
    np.savez_compressed(os.path.join(HERE, "sah_rows.npz"), **out)
    print({k: v.shape for k, v in out.items()})
