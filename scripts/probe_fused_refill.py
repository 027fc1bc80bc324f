"""dev probe: pt_tuning.refill (the shade block runs once refill / 64 of the wave's live lanes wait) on the fused kernel, library shapes, 1080p."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H = 1920, 1080
for K in (16, 4, 1):
    row = []
    for rep in range(2):
        for r in (-1, 24, 32, 36, 40, 44, 48, 56):
            ctx.set_tuning(refill=r)
            film = pt.Film(ctx, W, H)
            p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED)
            pt.render(sc, film, p)
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
            row.append(f"r{r} {statistics.median(ts) * 1e3:.3f}")
            film.close()
        row.append("|")
    print(f"K {K}: " + "  ".join(row), flush=True)
