"""dev probe: config C4 (10 000 instances), ms per frame by frames per call: the fused two-level kernel against the wavefront pipeline (library shapes)."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
sc.set_instances(pt.cornell_grid_instances())
W, H = 1920, 1080
for K in (1, 2, 4, 8, 16):
    row = []
    for name, pl in (("fused", pt.PIPELINE_FUSED), ("wavefront", pt.PIPELINE_WAVEFRONT)):
        film = pt.Film(ctx, W, H)
        p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pl)
        pt.render(sc, film, p)
        ts = []
        for _ in range(5):
            ctx.reset_stats()
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        row.append(f"{name} {statistics.median(ts) * 1e3 / K:.3f} ms/frame ({st.rays / statistics.median(ts) / 1e9:.2f} Grays/s, groups {st.sample_groups}, {st.workspace_bytes / 2**30:.1f} GB)")
        film.close()
    print(f"C4 K {K}: " + "  |  ".join(row), flush=True)
