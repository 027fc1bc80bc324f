"""dev probe: what a fused call costs when it has next to no work (8 x 8 .. 256 x 256 films), per shape."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
for (W, H) in ((8, 8), (64, 64), (128, 128), (256, 256)):
    row = []
    for name, shape in (("g1", dict(sample_groups=1)), ("g32", dict(sample_groups=32)), ("auto", dict())):
        film = pt.Film(ctx, W, H)
        p = pt.default_params(frame=0, frame_count=1, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, **shape)
        pt.render(sc, film, p)
        ts = []
        for _ in range(30):
            ctx.reset_stats()
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        row.append(f"{name} wall {statistics.median(ts) * 1e6:.0f} us (device {st.ms_total * 1e3:.0f} us, rays {st.rays})")
        film.close()
    print(f"{W}x{H}: " + "  ".join(row), flush=True)
