"""dev probe: the fused hand-out order with the scene-box tiles first (pt_tuning.fused_subject) against centre first only, per shape:
ms per call at 1080p, 32 spp, depth 8 for K frames per call; g1 = plain one group, g32 = every sample a slot, S = head + tail with S tail samples."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H, spp = 1920, 1080, 32
Ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3, 4, 8, 16]
ref = {}
for K in Ks:
    for subj in (0, 1):
        row = []
        for name, tune, shape in [("g1", 0, dict(sample_groups=1)), ("g32", 0, dict(sample_groups=32))] + [(f"S{S}", S, {}) for S in (2, 4, 8, 12, 16, 20)]:
            if K > 4 and name in ("g32", "S16", "S20"):
                continue
            ctx.set_tuning(fused_tail=tune, fused_subject=subj)
            film = pt.Film(ctx, W, H)
            p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=spp, max_depth=8, pipeline=pt.PIPELINE_FUSED, **shape)
            pt.render(sc, film, p)
            ts = []
            for _ in range(7):
                film.clear(); ctx.reset_stats()
                t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
            st = ctx.stats()
            img = film.read_f32().tobytes()
            ref.setdefault(K, (img, st.rays))
            ok = img == ref[K][0] and st.rays == ref[K][1]
            row.append(f"{name} {statistics.median(ts) * 1e3:.3f}{'' if ok else ' MISMATCH'}")
            film.close()
        print(f"K {K} subject-first {subj}: ms per call: " + "  ".join(row), flush=True)
