"""dev tool: randomized render parity -- random film sizes (ragged), spp, depth, frame ranges, frames in flight,
sample groups, tail samples, the cull and the hand-out order on / off, rank/world splits, extend variants, both pipelines, cameras -- GPU film vs the oracle's, bit for bit."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
from oracle import pt_oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = pt.Context(0)
arrays = pt.load_obj(pt.ASSET_CORNELL)
sc, osc = pt.Scene(ctx, *arrays), orc.Scene(*arrays)
bad = 0
t0 = time.time()
for k in range(N):
    rng = np.random.default_rng(SEED0 + k)
    w, h = int(rng.integers(1, 90)), int(rng.integers(1, 60))
    spp = int(rng.choice([1, 2, 3, 5, 8, 12, 32]))
    depth = int(rng.choice([1, 2, 4, 8, 13]))
    f0, nf = int(rng.choice([0, 0, 1, 7])), int(rng.integers(1, 5))
    kw = dict(width=w, height=h, spp_per_frame=spp, max_depth=depth)
    if rng.random() < 0.3:
        kw.update(cam_origin=tuple(float(x) for x in rng.uniform(-0.5, 0.5, 3) + np.array([0, -1, 4.0])),
                  env=tuple(float(x) for x in rng.uniform(0, 1, 3)))
    # a third of the configurations: a two-level scene (2 .. 40 rotated + scaled instances of the box; both pipelines have a kernel for it)
    n_inst = int(rng.integers(2, 41)) if rng.random() < 0.33 else 0
    inst = np.zeros((n_inst, 3, 4), np.float32)
    for i in range(n_inst):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        inst[i, :, :3] = (q * rng.uniform(0.2, 0.6)).astype(np.float32)
        inst[i, :, 3] = rng.uniform(-1.5, 1.5, 3).astype(np.float32) + np.float32([0, -1, 0])
    sc.set_instances(inst)
    osc.set_instances(inst)
    # oracle: frames f0 .. f0+nf-1 blended onto a film that already holds frames 0 .. f0-1
    film_o = None
    for fr in range(0, f0 + nf):
        img, _, _, _ = osc.render_frame(orc.default_params(frame=fr, **kw))
        if film_o is None:
            film_o = np.zeros_like(img)
        orc.accumulate_f32(film_o, img, fr)
    world = int(rng.choice([1, 1, 2, 3, 5]))
    total = np.zeros((h, w, 3), np.float32)
    for rank in range(world):
        film = pt.Film(ctx, w, h)
        gk = dict(kw, rank=rank, world=world, frames_in_flight=int(rng.choice([0, 1, 2, 5])), sample_groups=int(rng.choice([0, 1, 2, 3, spp])),
                  extend=int(rng.choice([pt.EXTEND_AUTO, pt.EXTEND_LDS, pt.EXTEND_HBM, pt.EXTEND_HBM8])))
        if n_inst:
            gk.update(extend=pt.EXTEND_AUTO)   # (one two-level kernel per scene class)
        r = rng.random()
        if r < 0.35:     # the fused single-kernel pipeline (LDS scenes; the compact pair-leaf tree / the two-level one)
            gk.update(pipeline=pt.PIPELINE_FUSED, extend=pt.EXTEND_AUTO)
        elif r < 0.7:    # the wavefront queues, named
            gk.update(pipeline=pt.PIPELINE_WAVEFRONT)
        else:            # PT_PIPELINE_AUTO (what pt_params_default returns): fused where the call allows it, else wavefront
            gk.update(pipeline=pt.PIPELINE_AUTO)
        # the fused kernel's head + tail slots (the library's rule, never, S); the cull of pixels that cannot see the scene and the subject-first order (on, on, off)
        ctx.set_tuning(fused_tail=int(rng.choice([-1, -1, 0, 1, 2, 3, 7])), cull=int(rng.choice([-1, 1, 0])), fused_subject=int(rng.choice([-1, 1, 0])))
        if f0:
            pt.render(sc, film, pt.default_params(frame=0, frame_count=f0, **gk))
        pt.render(sc, film, pt.default_params(frame=f0, frame_count=nf, **gk))
        total += film.read_f32()          # x + 0 == x: shards sum exactly
        film.close()
    if total.tobytes() != film_o.tobytes():
        bad += 1
        print("MISMATCH", k, kw, "instances", n_inst, "world", world, "max abs diff", float(np.abs(total - film_o).max()))
print(f"render fuzz: {N} configurations, mismatches: {bad}; {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
