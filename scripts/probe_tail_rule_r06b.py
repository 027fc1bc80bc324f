import importlib, os, statistics, sys, time
sys.path.insert(0, "/root/repo")
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
for (W,H,K) in ((1920,1080,1),(1920,1080,2),(1920,1080,3),(1024,1024,1),(1280,720,1)):
    row = []
    for S in (-1, 0, 8, 12, 14, 16, 18, 20, 24):
        old = ctx.set_tuning(fused_tail=S)
        film = pt.Film(ctx, W, H)
        p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED)
        pt.render(sc, film, p)
        ts = []
        for _ in range(9):
            film.clear(); ctx.reset_stats()
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        row.append(f"S{S if S >= 0 else 'dflt->' + str(st.tail_samples)} {statistics.median(ts) * 1e3:.3f}")
        film.close()
        ctx.set_tuning(**old)
    print(f"{W}x{H} K {K}: " + " | ".join(row), flush=True)
