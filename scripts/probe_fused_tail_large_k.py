"""dev probe: one or two tail samples on launches of many frames (with the cull the border's cheap slots no longer cover a launch's end)."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H = 1920, 1080
for K in (6, 8, 12, 16, 20, 32):
    row, ref = [], None
    for S in (0, 1, 2, 3):
        ctx.set_tuning(fused_tail=S)
        film = pt.Film(ctx, W, H)
        p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED)
        pt.render(sc, film, p)
        ts = []
        for _ in range(5):
            ctx.reset_stats()
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        img = film.read_f32().tobytes()
        ref = ref or img
        row.append(f"S{S} {statistics.median(ts) * 1e3:.3f}{'' if img == ref else ' MISMATCH'} ({st.workspace_bytes / 2**30:.1f} GB)")
        film.close()
    print(f"K {K}: ms per call: " + "  ".join(row), flush=True)
