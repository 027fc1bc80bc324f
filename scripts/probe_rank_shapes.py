"""dev probe: one rank of world 8 / 4 at C3's size (1080p, 32 spp, 32 frames) and at 16 frames through the fused pipeline: the library's shape against hand-picked
tail samples and frames in flight -- what the shard efficiency of the emulation (probe_shard_efficiency.py) is made of."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
for world, K in ((8, 32), (8, 16), (4, 32), (4, 16)):
    row = []
    for name, tune, extra in (("default", {}, {}), ("tail0", dict(fused_tail=0), {}), ("tail2", dict(fused_tail=2), {}), ("tail4", dict(fused_tail=4), {}), ("tail6", dict(fused_tail=6), {}),
                              ("tail8", dict(fused_tail=8), {}), ("tail12", dict(fused_tail=12), {}), ("tail16", dict(fused_tail=16), {}),
                              ("fif/2", {}, dict(frames_in_flight=K // 2)), ("fif/2 tail8", dict(fused_tail=8), dict(frames_in_flight=K // 2))):
        old = ctx.set_tuning(**tune) if tune else {}
        film = pt.Film(ctx, 1920, 1080)
        p = pt.default_params(frame=0, frame_count=K, width=1920, height=1080, spp_per_frame=32, max_depth=8, rank=1, world=world, pipeline=pt.PIPELINE_FUSED, **extra)
        pt.render(sc, film, p)
        ts = []
        for _ in range(5):
            film.clear(); ctx.reset_stats()
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        row.append(f"{name} {statistics.median(ts) * 1e3:.3f} (S{st.tail_samples} f{st.frames_in_flight})")
        film.close()
        if old:
            ctx.set_tuning(**old)
    print(f"world {world} K {K}: " + " | ".join(row), flush=True)
