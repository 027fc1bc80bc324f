"""dev check: PT_PIPELINE_AUTO on edge inputs against the oracle, bit for bit: one triangle, scenes at the borders of the fused kernel's class, 1 spp / depth 1,
thousands of spp on a tiny film, more ranks than tiles, 1 x 1 films, a single frame far into a progression."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
from oracle import pt_oracle as orc
ctx = pt.Context(0)
bad = 0
def soup(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
    v = (c + rng.uniform(-0.3, 0.3, (n, 3, 3)).astype(np.float32)).astype(np.float32)
    v[..., 1] -= 1.0
    f = rng.uniform(0, 1, (n, 6)).astype(np.float32); f[:, 3:] *= (rng.uniform(0, 1, (n, 1)) < 0.2)
    return v.reshape(-1), np.arange(3 * n, dtype=np.uint32), f.reshape(-1).astype(np.float32)
def check(name, arrays, frames=(0, 1), world=1, **kw):
    global bad
    sc, osc = pt.Scene(ctx, *arrays), orc.Scene(*arrays)
    w, h = kw["width"], kw["height"]
    film_o = np.zeros((h, w, 3), np.float32); rays_o = 0
    for fr in range(frames[0] + frames[1]):
        img, r, _, _ = osc.render_frame(orc.default_params(frame=fr, **kw))
        orc.accumulate_f32(film_o, img, fr)
        if fr >= frames[0]:
            rays_o += r
    total = np.zeros_like(film_o); rays = 0; pipes = set()
    for rank in range(world):
        film = pt.Film(ctx, w, h)
        if frames[0]:
            pt.render(sc, film, pt.default_params(frame=0, frame_count=frames[0], rank=rank, world=world, **kw))
        ctx.reset_stats()
        pt.render(sc, film, pt.default_params(frame=frames[0], frame_count=frames[1], rank=rank, world=world, **kw))
        st = ctx.stats(); rays += st.rays; pipes.add(pt.PIPELINE_NAMES[st.pipeline])
        total += film.read_f32(); film.close()
    ok = total.tobytes() == film_o.tobytes() and rays == rays_o
    bad += not ok
    print(f"{name:46s} {'ok ' if ok else 'MISMATCH'} pipelines {sorted(pipes)} rays {rays} (oracle {rays_o})", flush=True)
    sc.close()
cornell = pt.load_obj(pt.ASSET_CORNELL)
one = (np.float32([-1, -2, 0, 1, -2, 0, 0, 0, 0]), np.uint32([0, 1, 2]), np.float32([.5, .5, .5, 1, 2, 3]))
check("one emissive triangle", one, width=40, height=24, spp_per_frame=4, max_depth=8)
check("Cornell, 1 spp, depth 1", cornell, width=64, height=40, spp_per_frame=1, max_depth=1)
check("Cornell, 1 x 1 film", cornell, width=1, height=1, spp_per_frame=32, max_depth=8)
check("Cornell, 4096 spp on 4 x 3", cornell, frames=(0, 1), width=4, height=3, spp_per_frame=4096, max_depth=8)
check("Cornell, depth 40, 7 spp", cornell, width=33, height=17, spp_per_frame=7, max_depth=40)
check("Cornell, frame 500 alone after 0..499", cornell, frames=(500, 1), width=16, height=16, spp_per_frame=2, max_depth=4)
check("Cornell, world 37 on a 24 x 16 film", cornell, world=37, width=24, height=16, spp_per_frame=3, max_depth=8)
check("soup of 204 triangles (last fused size)", soup(204, 3), width=48, height=32, spp_per_frame=3, max_depth=6)
check("soup of 205 triangles (first wavefront size)", soup(205, 3), width=48, height=32, spp_per_frame=3, max_depth=6)
check("soup of 2047 triangles", soup(2047, 4), width=48, height=32, spp_per_frame=2, max_depth=6)
check("soup of 2049 triangles", soup(2049, 4), width=48, height=32, spp_per_frame=2, max_depth=6)
print("edge cases:", "all equal" if not bad else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
