"""dev probe: explicit shapes (all groups, plain, tails) where the slots per lane of the grid are few or many."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
spp = 32
for (W, H, K) in ((256, 256, 1), (256, 256, 4), (640, 360, 1), (640, 360, 4), (960, 540, 2), (1280, 720, 1), (1280, 720, 2), (1920, 1080, 1), (3840, 2160, 1), (3840, 2160, 2), (2560, 1440, 2)):
    row, ref = [], None
    for name, tune, shape in [("g32", 0, dict(sample_groups=32)), ("g1", 0, dict(sample_groups=1))] + [(f"S{S}", S, {}) for S in (4, 8, 12, 16, 20, 24, 28)]:
        ctx.set_tuning(fused_tail=tune)
        film = pt.Film(ctx, W, H)
        p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=spp, max_depth=8, pipeline=pt.PIPELINE_FUSED, **shape)
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
        row.append(f"{name} {statistics.median(ts) * 1e3:.3f}{'' if ok else ' MISMATCH'}")
        film.close()
    print(f"{W}x{H} K {K} (x2 = {W * H * K * 2 // 393216}): ms per call: " + "  ".join(row), flush=True)
