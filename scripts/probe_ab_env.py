"""dev probe: ms per call of the fused pipeline at 1080p for K = 16, 4, 1 with pt_tuning set from argv ("refill=36 ..."); one process per build / setting
for an A/B on one box."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
for a in sys.argv[1:]:
    k, v = a.split("=")
    ctx.set_tuning(**{k: int(v)})
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H = 1920, 1080
out = []
for K in (16, 4, 1):
    film = pt.Film(ctx, W, H)
    p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED)
    pt.render(sc, film, p)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
    out.append(f"K {K}: {statistics.median(ts) * 1e3:.3f} ms")
    film.close()
print(os.environ.get("PT_LIB_AMD", "product"), " ".join(sys.argv[1:]), " | ".join(out), flush=True)
