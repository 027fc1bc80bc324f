"""dev probe: the fused pipeline's HEAD + TAIL shape (pt_tuning.fused_tail = S one-sample tail slots per pixel behind a head slot of 32 - S samples) at K = 1 / 2 / 4 / 8
frames per call: ms per frame by S (0 = the plain shape), film compared with the plain one bit for bit."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
W, H = 1920, 1080
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
for K in (1, 2, 4, 8):
    row, ref = [], None
    for S in (0, 2, 4, 8, 12, 16, 24):
        ctx.set_tuning(fused_tail=S)
        film = pt.Film(ctx, W, H)
        p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED)
        pt.render(sc, film, p)
        ts = []
        for _ in range(7):
            film.clear(); ctx.reset_stats()
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        img = film.read_f32().tobytes()
        if ref is None:
            ref = (img, st.rays)
        ok = img == ref[0] and st.rays == ref[1]
        row.append(f"S{S} {statistics.median(ts) * 1e3 / K:.3f}{'' if ok else ' MISMATCH'} (g{st.sample_groups} t{st.tail_samples} {st.workspace_bytes / 2**30:.1f}GB)")
        film.close()
    print(f"K {K}: ms per frame: " + "  ".join(row), flush=True)
