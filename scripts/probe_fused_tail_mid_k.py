"""dev probe: tail samples at 5 .. 7 frames per call and on a rank of world 4 at 20 frames (10 .. 19 walked slots per lane), ms per call."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H = 1920, 1080
for K, world in ((5, 1), (6, 1), (7, 1), (20, 4), (16, 4), (20, 2)):
    row = []
    for rep in range(2):
        for S in (0, 2, 3, 4, 6):
            ctx.set_tuning(fused_tail=S)
            film = pt.Film(ctx, W, H)
            p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, rank=1 if world > 1 else 0, world=world)
            pt.render(sc, film, p)
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
            row.append(f"S{S} {statistics.median(ts) * 1e3:.3f}")
            film.close()
        row.append("|")
    print(f"K {K} world {world}: " + "  ".join(row), flush=True)
