"""dev probe: pt_tuning.fused_subject on views whose subject is not in the middle of the image (the box pushed to the left / a corner, a far box),
where centre-first alone hands cheap tiles out early and the subject's last: ms per call, library shapes, 1080p, 32 spp, depth 8."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H, spp = 1920, 1080, 32
views = {"reference": {}, "box at the left": dict(cam_origin=(1.1, -1.0, 5.0), cam_target=(1.1, -1.0, 2.0)),
         "box in a corner": dict(cam_origin=(1.0, -0.2, 5.0), cam_target=(1.0, -0.2, 2.0)), "far box": dict(cam_origin=(0.0, -1.0, 9.0), cam_target=(0.0, -1.0, 6.0))}
for vname, cam in views.items():
    for K in (1, 4, 16):
        row = []
        ref = None
        for subj in (0, 1):
            for name, tune, shape in (("plain", 0, dict(sample_groups=1)), ("library", -1, {})):
                ctx.set_tuning(fused_tail=tune, fused_subject=subj)
                film = pt.Film(ctx, W, H)
                p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=spp, max_depth=8, pipeline=pt.PIPELINE_FUSED, **cam, **shape)
                pt.render(sc, film, p)
                ts = []
                for _ in range(5):
                    film.clear(); ctx.reset_stats()
                    t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
                st = ctx.stats()
                img = film.read_f32().tobytes()
                ref = ref or (img, st.rays)
                ok = img == ref[0] and st.rays == ref[1]
                row.append(f"subject {subj} {name} {statistics.median(ts) * 1e3:.3f}{'' if ok else ' MISMATCH'}")
                film.close()
        print(f"{vname}, K {K} ({ref[1] / K / 1e6:.1f} Mrays per frame): " + "  ".join(row), flush=True)
