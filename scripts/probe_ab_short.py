"""dev probe: library shapes at K = 1, 2, 3 and a rank of world 8 at 16 frames (the head + tail shapes), once per library build (PT_LIB_AMD)."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H = 1920, 1080
out = []
for K, world in ((1, 1), (2, 1), (3, 1), (16, 8), (16, 1)):
    film = pt.Film(ctx, W, H)
    p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, rank=0, world=world)
    pt.render(sc, film, p)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
    out.append(f"K {K}/w{world}: {statistics.median(ts) * 1e3:.3f}")
    film.close()
print(os.environ.get("PT_LIB_AMD", "product"), " | ".join(out), flush=True)
