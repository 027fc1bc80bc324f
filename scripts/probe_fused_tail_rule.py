"""dev probe: the head + tail rule against S = 0 and neighbours at other film sizes and spp (render.hip fused_tail_samples)."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
for (W, H, spp, K) in ((1280, 720, 32, 1), (1280, 720, 32, 2), (1280, 720, 32, 4), (960, 540, 32, 1), (960, 540, 32, 4), (2560, 1440, 32, 1), (3840, 2160, 32, 1),
                       (1920, 1080, 64, 1), (1920, 1080, 64, 2), (1920, 1080, 16, 1), (1920, 1080, 16, 2), (1920, 1080, 8, 2), (1920, 1080, 128, 1)):
    row, ref = [], None
    for S in (-1, 0, spp // 4, spp * 3 // 8, spp // 2, spp * 5 // 8):
        ctx.set_tuning(fused_tail=S)
        film = pt.Film(ctx, W, H)
        p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=spp, max_depth=8, pipeline=pt.PIPELINE_FUSED)
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
        row.append(f"S{S}{'->' + str(st.tail_samples) + '/g' + str(st.sample_groups) if S < 0 else ''} {statistics.median(ts) * 1e3:.3f}{'' if ok else ' MISMATCH'}")
        film.close()
    print(f"{W}x{H} spp {spp} K {K}: ms per call: " + "  ".join(row), flush=True)
