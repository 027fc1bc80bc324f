"""dev probe (round 6): the head + tail rule's upper buckets re-fitted on the faster kernel: 1080p Cornell, fused, K = 2 .. 8 frames per call (6.1 .. 24.4 walked slots per
lane), ms per call by tail samples S (default = the library's rule)."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
for K in (2, 3, 4, 5, 6, 7, 8):
    row = []
    for S in (-1, 0, 2, 4, 6, 8, 10, 12):
        old = ctx.set_tuning(fused_tail=S)
        film = pt.Film(ctx, 1920, 1080)
        p = pt.default_params(frame=0, frame_count=K, width=1920, height=1080, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED)
        pt.render(sc, film, p)
        ts = []
        for _ in range(7):
            film.clear(); ctx.reset_stats()
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        row.append(f"S{S if S >= 0 else 'dflt->' + str(st.tail_samples)} {statistics.median(ts) * 1e3:.3f}")
        film.close()
        ctx.set_tuning(**old)
    print(f"K {K} ({3.05 * K:.1f} walked slots per lane): " + " | ".join(row), flush=True)
