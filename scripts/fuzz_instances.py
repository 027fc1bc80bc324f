"""dev tool: randomized two-level (instanced) closest-hit parity: random base scenes, random instance sets
(rotations, non-uniform and mirrored scales 0.01..10, translations up to 100 scene sizes, duplicated instances)
and rays; GPU k_extend_inst vs the oracle (TLAS walk and brute force over every (instance, triangle))."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
from oracle import pt_oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = pt.Context(0)
bad = 0
ndeg = 0
n_refused = 0
t0 = time.time()
def rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q); a, b, c, d = q
    return np.array([[a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c)], [2*(b*c+a*d), a*a-b*b+c*c-d*d, 2*(c*d-a*b)], [2*(b*d-a*c), 2*(c*d+a*b), a*a-b*b-c*c+d*d]])
for k in range(N):
    rng = np.random.default_rng(SEED0 + k)
    n = int(rng.choice([1, 2, 12, 36, 150, 600, 3000]))
    ctr = rng.uniform(-1, 1, (n, 1, 3)); spread = float(rng.choice([0.05, 0.3]))
    tri = (ctr + rng.uniform(-spread, spread, (n, 3, 3))).astype(np.float32)
    v = tri.reshape(-1); idx = np.arange(3 * n, dtype=np.uint32); f = rng.uniform(0, 1, 6 * n).astype(np.float32)
    ni = int(rng.choice([1, 2, 5, 40, 300]))
    xf = np.zeros((ni, 3, 4), np.float32)
    field = float(rng.choice([3.0, 30.0, 300.0]))
    for j in range(ni):
        s = rng.choice([0.01, 0.3, 1.0, 1.0, 10.0]) * rng.uniform(0.5, 1.5, 3)
        if rng.random() < 0.2: s[int(rng.integers(3))] *= -1          # mirrored
        M = rot(rng) @ np.diag(s) if rng.random() < 0.8 else np.diag(s)
        xf[j, :, :3] = M; xf[j, :, 3] = rng.uniform(-field, field, 3)
    if ni > 3 and rng.random() < 0.3: xf[ni // 2] = xf[0]                 # coincident instances: lowest instance id wins
    m = 20000
    pick = rng.integers(0, ni, m)
    local = tri[rng.integers(0, n, m)].mean(1) + rng.normal(size=(m, 3)) * 0.2
    tgt = np.einsum('mij,mj->mi', xf[pick, :, :3], local) + xf[pick, :, 3]
    org = tgt + rng.normal(size=(m, 3)) * rng.choice([0.5, 5.0, 50.0], (m, 1)) * rng.uniform(0.2, 2.0, (m, 1))
    d = tgt - org
    ax = rng.random(m) < 0.15
    d[ax] = 0; d[ax, rng.integers(0, 3, ax.sum())] = rng.choice([-1.0, 1.0], ax.sum())
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], axis=1).astype(np.float32)
    tmin, tmax = float(rng.choice([1e-3, 0.0])), float(rng.choice([1e4, 1e9]))
    osc = orc.Scene(v, idx, f); osc.set_instances(xf)
    want, _ = osc.trace(rays, tmin=tmin, tmax=tmax, mode=1)
    brute = osc.trace(rays, tmin=tmin, tmax=tmax, mode=0)[0] if n * ni <= 200000 else want
    agree = (brute.view(np.uint8).reshape(m, -1) == want.view(np.uint8).reshape(m, -1)).all(axis=1)
    ndeg += int((~agree).sum())
    gs = pt.Scene(ctx, v, idx, f)
    if rng.random() < 0.5: gs.set_bvh_quality(pt.BVH_PREFER_FAST_BUILD)
    gs.set_instances(xf)
    got = gs.trace(rays, tmin=tmin, tmax=tmax)
    gb = got.view(np.uint8).reshape(m, -1)
    ok = np.where(agree, (gb == want.view(np.uint8).reshape(m, -1)).all(axis=1),
                  (gb == want.view(np.uint8).reshape(m, -1)).all(axis=1) | (gb == brute.view(np.uint8).reshape(m, -1)).all(axis=1))
    if not ok.all():
        bad += 1
        dd = np.nonzero(~ok)[0]
        print("MISMATCH", k, "n", n, "instances", ni, len(dd), "rays; first", got[dd[:2]], want[dd[:2]])
    # the same scene through both render pipelines (k_shade<INST> / k_fused_inst) against the oracle's film: camera on instance 0
    if tmin > 0.0:
        c0 = xf[0, :, 3].astype(np.float64)
        eye = c0 + rng.normal(size=3) * float(np.abs(xf[0, :, :3]).max()) * 3.0
        kw = dict(width=48, height=32, spp_per_frame=2, max_depth=5, tmin=tmin, tmax=tmax, cam_origin=tuple(float(x) for x in eye),
                  cam_target=tuple(float(x) for x in c0))
        ofilm = None
        for fr in range(2):
            img, _, _, _ = osc.render_frame(orc.default_params(frame=fr, **kw))
            if ofilm is None:
                ofilm = np.zeros_like(img)
            orc.accumulate_f32(ofilm, img, fr)
        for pipe in (pt.PIPELINE_WAVEFRONT, pt.PIPELINE_FUSED):
            film = pt.Film(ctx, 48, 32)
            try:
                pt.render(gs, film, pt.default_params(frame=0, frame_count=2, pipeline=pipe, **kw))
                if film.read_f32().tobytes() != ofilm.tobytes():
                    bad += 1
                    print("RENDER MISMATCH", k, "pipeline", pipe, "n", n, "instances", ni)
            except pt.PtError as e:   # (the fused pipeline takes the fp16 two-level kernel's scenes only)
                assert pipe == pt.PIPELINE_FUSED and e.status == 5, str(e)
                n_refused += 1
            film.close()
    gs.close()
print(f"fused refusals (scene outside k_fused_inst's class): {n_refused}")
print(f"instance fuzz: {N} scenes, mismatching scenes: {bad}; rays where oracle brute force and TLAS walk differ: {ndeg}; {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
