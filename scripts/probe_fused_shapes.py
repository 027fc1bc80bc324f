"""dev probe: the fused pipeline over the shapes that matter -- K = 16 / 2 / 1 in one call (one group forced, and the library's own shape), a rank
of world 8 at K = 32 and 16, config C4 at K = 8 -- median of 5 blocking calls each.  For A/B runs of variant builds (PT_LIB_AMD)."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
W, H = 1920, 1080
def run(sc, K, groups, world=1, reps=5):
    film = pt.Film(ctx, W, H)
    p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, rank=0, world=world,
                          pipeline=pt.PIPELINE_FUSED, sample_groups=groups)
    pt.render(sc, film, p)
    ts = []
    for _ in range(reps):
        film.clear(); ctx.reset_stats()
        t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
    st = ctx.stats()
    film.close()
    t = statistics.median(ts)
    return f"{st.rays / t / 1e6:8.0f} Mrays/s {t * 1e3 / K:7.3f} ms/frame (fif {st.frames_in_flight} x {st.sample_groups} groups)"
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
tag = os.environ.get("PT_LIB_AMD", "product").split("/")[-2] if os.environ.get("PT_LIB_AMD") else "product"
for K, g in ((16, 1), (4, 1), (2, 1), (2, 0), (1, 1), (1, 0)):
    print(tag, f"C2 K {K:2d} groups {'auto' if g == 0 else g}:", run(sc, K, g), flush=True)
for K in (32, 16):
    print(tag, f"C2 K {K:2d} rank 0 of world 8:", run(sc, K, 0, world=8), flush=True)
sc.close()
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
sc.set_instances(pt.cornell_grid_instances())
print(tag, "C4 K  8 groups auto:", run(sc, 8, 0, reps=3), flush=True)
