"""dev probe: what the sample-group form of the fused kernel costs -- one blocking 1080p frame (and K = 2) with 1 / 8 / 32 groups: wall time of
pt_render and the kernel's own duration (the rest is k_resolve replaying the logs + launch gaps).  For variant builds (PT_LIB_AMD)."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
W, H = 1920, 1080
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
tag = os.environ.get("PT_LIB_AMD", "x/product/y").split("/")[-2]
for K, G in ((1, 1), (1, 8), (1, 32), (2, 1), (2, 16), (4, 8)):
    film = pt.Film(ctx, W, H)
    p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, sample_groups=G,
                          frames_in_flight=K, flags=pt.FLAG_PROFILE)
    pt.render(sc, film, p)
    rows = []
    for _ in range(7):
        film.clear(); ctx.reset_stats()
        t0 = time.perf_counter(); pt.render(sc, film, p); w = (time.perf_counter() - t0) * 1e3
        st = ctx.stats()
        rows.append((w, st.ms_extend, st.ms_total))
    rows.sort()
    w, k, tot = rows[len(rows) // 2]
    print(f"{tag} K {K} G {G:2d}: wall {w / K:.3f} ms/frame, k_fused {k / K:.3f}, device {tot / K:.3f}; {st.rays / w / 1e3:.0f} Mrays/s", flush=True)
    film.close()
