"""dev tool: PT_PIPELINE_FUSED on instanced scenes against the wavefront pipeline (itself pinned to the oracle by the tests) --
film bits and ray counts over instance sets, film shapes, sample groups, frames in flight, shards and tuning knobs."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")


def random_instances(n, seed):
    rng = np.random.default_rng(seed)
    m = np.zeros((n, 3, 4), np.float32)
    for k in range(n):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        m[k, :, :3] = (q * rng.uniform(0.2, 0.6)).astype(np.float32)
        m[k, :, 3] = rng.uniform(-1.5, 1.5, 3).astype(np.float32) + np.float32([0, -1, 0])
    return m


ctx = pt.Context(0)
arrays = pt.load_obj(pt.ASSET_CORNELL)
bad = 0
t0 = time.time()
sets = [("rand1", random_instances(1, 1), {}), ("rand5", random_instances(5, 3), {}), ("rand60", random_instances(60, 4), {}),
        ("rand1500", random_instances(1500, 5), {}), ("grid", pt.cornell_grid_instances(), {}),
        ("grid_close", pt.cornell_grid_instances(), dict(cam_origin=(-0.88, -1.9, 0.5), cam_target=(-0.88, -1.9, 0.0)))]
for name, inst, cam in sets:
    sc = pt.Scene(ctx, *arrays)
    sc.set_instances(inst)
    for (w, h, spp, depth, nf) in ((80, 64, 3, 6, 2), (131, 77, 8, 8, 3), (320, 200, 4, 13, 1)):
        kw = dict(width=w, height=h, spp_per_frame=spp, max_depth=depth, frame=0, frame_count=nf, **cam)
        ref = pt.Film(ctx, w, h)
        ctx.reset_stats()
        pt.render(sc, ref, pt.default_params(**kw))
        want, rays_want = ref.read_f32(), ctx.stats().rays
        ref.close()
        for extra in (dict(), dict(sample_groups=spp), dict(sample_groups=2 if spp % 2 == 0 else 1, frames_in_flight=1), dict(frames_in_flight=2)):
            film = pt.Film(ctx, w, h)
            ctx.reset_stats()
            try:
                pt.render(sc, film, pt.default_params(pipeline=pt.PIPELINE_FUSED, **kw, **extra))
            except pt.PtError as e:   # (one instance: no fp16 TLAS is built, the general two-level kernel walks it; the fused pipeline has no form of that)
                assert len(inst) == 1 and e.status == 5, (name, str(e))
                film.close()
                continue
            got, rays = film.read_f32(), ctx.stats().rays
            film.close()
            ok = got.tobytes() == want.tobytes() and rays == rays_want
            if not ok:
                bad += 1
                print("MISMATCH", name, (w, h, spp, depth, nf), extra, "rays", rays, rays_want, "max abs", float(np.abs(got - want).max()),
                      "pixels differing", int((got != want).any(axis=2).sum()))
    sc.close()
print(f"fused x instances: mismatches {bad}; {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
