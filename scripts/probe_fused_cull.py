"""dev probe: pt_tuning.cull (slots of pixels that cannot see the scene are finished without a walk) off / on, per shape: ms per call at
1080p, 32 spp, depth 8, K frames per call; g1 = one group, g32 = every sample a slot, S = head + tail with S tail samples, lib = the library's rule."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H, spp = 1920, 1080, 32
Ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3, 4, 8, 16]
ref = {}
for K in Ks:
    for cull in (0, 1):
        row = []
        shapes = [("g1", 0, dict(sample_groups=1)), ("g32", 0, dict(sample_groups=32))] + [(f"S{S}", S, {}) for S in (2, 4, 8, 12, 16, 20)] + [("lib", -1, {})]
        for name, tune, shape in shapes:
            if K > 4 and name in ("g32", "S16", "S20"):
                continue
            ctx.set_tuning(fused_tail=tune, cull=cull)
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
        print(f"K {K} cull {cull} (rays {st.rays}, culled {st.rays_culled}): ms per call: " + "  ".join(row), flush=True)
