"""dev probe: per-rank efficiency of the pixel-tile sharding, emulated on ONE GPU: time rank 0's share of a
world-N split (no collectives) against 1/N of the full image.  K frames as bench.py's default."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0); sc = pt.Scene.from_obj(ctx); film = pt.Film(ctx, 1920, 1080)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
base = None
for world in (1, 2, 4, 8):
    kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8, rank=0, world=world)
    p = pt.default_params(frame=0, frame_count=K, **kw)
    pt.render_prepare(sc, film, p)
    pt.render(sc, film, pt.default_params(frame=0, frame_count=2, **kw))       # warm-up
    film.clear(); ctx.reset_stats()
    t0 = time.perf_counter(); pt.render(sc, film, p); dt = time.perf_counter() - t0
    st = ctx.stats()
    if world == 1: base = dt
    print(f"world {world}: rank-0 time {dt*1e3:8.2f} ms  rays {st.rays:>11d}  {st.rays/dt/1e6:9.1f} Mrays/s per rank "
          f"-> ideal {base/world*1e3:7.2f} ms, efficiency {base/world/dt*100:5.1f} %  (frames in flight {st.frames_in_flight}, groups {st.sample_groups})")
