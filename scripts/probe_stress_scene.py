#!/usr/bin/env python3
"""dev tool (GPU box): what ePreferFastTrace is worth above 2048 triangles.  The "teapot in a stadium" scene
(pth_make_stadium: primitive sizes over four orders of magnitude) and the uniform soup of config C5, each built with the
collapsed LBVH (PT_BVH_PREFER_FAST_BUILD) and with the PLOC rebuild (PT_BVH_PREFER_FAST_TRACE, the default): build time,
BVH4 node visits and triangle tests per ray of a 1080p render (device counters), Mrays/s of the same render without
counters, and that film and ray count agree bit for bit between the two trees.
usage: probe_stress_scene.py [floor_side sphere_seg]"""
import importlib, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
fs, seg = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 224)
ctx = pt.Context(0)
print("ploc_radius", ctx.tuning().ploc_radius, "(-1 = 8)")
W, H = 1920, 1080
out = []
for name, arrays, kw in ((f"stadium({fs},{seg})", pt.make_stadium(fs, seg), dict(spp_per_frame=8, max_depth=8)),
                         ("soup 1M (config C5)", pt.make_soup(1000000, 1), dict(spp_per_frame=8, max_depth=16))):
    sc = pt.Scene(ctx, *arrays)
    film = pt.Film(ctx, W, H)
    ref = None
    for qname, q in (("LBVH collapse (fast_build)", pt.BVH_PREFER_FAST_BUILD), ("PLOC rebuild (fast_trace)", pt.BVH_PREFER_FAST_TRACE)):
        t0 = time.perf_counter(); sc.set_bvh_quality(q); ctx.sync(); wall = time.perf_counter() - t0
        info = sc.info()
        common = dict(width=W, height=H, **kw)
        film.clear(); ctx.reset_stats()
        pt.render(sc, film, pt.default_params(frame=0, frame_count=2, flags=pt.FLAG_COUNT_VISITS, **common))
        c = ctx.stats()
        pt.render(sc, film, pt.default_params(frame=0, frame_count=2, **common))      # warm-up
        best = 0.0
        for _ in range(3):
            film.clear(); ctx.reset_stats()
            t0 = time.perf_counter(); pt.render(sc, film, pt.default_params(frame=0, frame_count=2, flags=pt.FLAG_PROFILE, **common)); dt = time.perf_counter() - t0
            st = ctx.stats()
            best = max(best, st.rays / dt / 1e6)
        img = film.read_f32().tobytes()
        rec = dict(scene=name, triangles=info.n_tris, tree=qname, builder=info.bvh4_builder, build_ms=round(info.build_ms, 2), rebuild_wall_ms=round(wall * 1e3, 1),
                   bvh4_nodes=info.n_wide_nodes, tree_area_lbvh=round(info.tree_area_lbvh, 1), tree_area_ploc=round(info.tree_area_ploc, 1), nodes_per_ray=round(c.nodes_visited / c.rays, 2), tris_per_ray=round(c.tris_tested / c.rays, 2),
                   mrays_per_s=round(best, 1), extend_ms=round(st.ms_extend, 2), shade_ms=round(st.ms_shade, 2), rays=st.rays)
        if ref is None:
            ref = (st.rays, img)
        else:
            rec["same_rays_and_film_as_the_lbvh"] = bool(ref[0] == st.rays and ref[1] == img)
        out.append(rec)
        print(json.dumps(rec), flush=True)
    film.close(); sc.close()
