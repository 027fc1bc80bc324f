"""dev probe: config C4 (10 000 instances, 1080p) through the wavefront pipeline (k_extend_inst16), ms per frame at 8 frames per call, no workspace budget; tuning from argv."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
ctx.set_tuning(mem_budget_mb=0)
for a in sys.argv[1:]:
    k, v = a.split("=")
    ctx.set_tuning(**{k: int(v)})
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
sc.set_instances(pt.cornell_grid_instances())
W, H = 1920, 1080
film = pt.Film(ctx, W, H)
p = pt.default_params(frame=0, frame_count=8, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_WAVEFRONT)
pt.render(sc, film, p)
ts = []
for _ in range(5):
    ctx.reset_stats()
    t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
st = ctx.stats()
print("wavefront", " ".join(sys.argv[1:]), f"K 8: {statistics.median(ts) * 1e3 / 8:.3f} ms/frame ({st.rays / statistics.median(ts) / 1e9:.2f} Grays/s)", flush=True)
