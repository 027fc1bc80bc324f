"""dev probe: the fused pipeline at K = 1 / 2 / 4 frames per call over sample-group counts (median of 7 blocking calls): which shape AUTO should take."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
W, H = 1920, 1080
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
tag = os.environ.get("PT_LIB_AMD", "x/product/y").split("/")[-2]
for K in (1, 2, 4):
    row = []
    for G in (1, 2, 4, 8, 16, 32):
        film = pt.Film(ctx, W, H)
        p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, sample_groups=G, frames_in_flight=K)
        pt.render(sc, film, p)
        ts = []
        for _ in range(7):
            film.clear()
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        row.append(f"G{G} {statistics.median(ts) * 1e3 / K:.3f}")
        film.close()
    print(tag, f"K {K}: ms per frame by groups:", "  ".join(row), flush=True)
