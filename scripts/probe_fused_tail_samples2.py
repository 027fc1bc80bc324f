"""dev probe: HEAD + TAIL (pt_tuning.fused_tail = S) at K = 3 on one device and on a rank of world 8 at K = 16 / 32: ms per frame by S."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
W, H = 1920, 1080
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
for K, world in ((3, 1), (16, 8), (32, 8), (8, 4), (2, 1), (1, 1)):
    row, ref = [], None
    for S in (0, 4, 8, 12, 16, 20):
        ctx.set_tuning(fused_tail=S)
        film = pt.Film(ctx, W, H)
        p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, rank=0, world=world)
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
        row.append(f"S{S} {statistics.median(ts) * 1e3:.3f}{'' if ok else ' MISMATCH'}")
        film.close()
    print(f"K {K} world {world}: ms per CALL: " + "  ".join(row), flush=True)
