"""dev probe: head + tail slots, S in single steps around the rule's choice (1080p, 32 spp, the cull on): ms per call, median of 9."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H = 1920, 1080
for K, Ss in ((1, (13, 14, 15, 16, 17, 18, 19)), (2, (9, 10, 11, 12, 13, 14, 15)), (3, (6, 7, 8, 9, 10)), (4, (3, 4, 5, 6, 7))):
    row = []
    for rep in range(2):
        for S in Ss:
            ctx.set_tuning(fused_tail=S)
            film = pt.Film(ctx, W, H)
            p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED)
            pt.render(sc, film, p)
            ts = []
            for _ in range(9):
                t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
            row.append(f"S{S} {statistics.median(ts) * 1e3:.3f}")
            film.close()
        row.append("|")
    print(f"K {K}: ms per call (two passes): " + "  ".join(row), flush=True)
