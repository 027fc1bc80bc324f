"""dev probe: where a fused launch's time goes when the job is small (a rank of world N): wall clock of pt_render, the device time between
its two events, the kernel's own duration (PT_FLAG_PROFILE), for K frames of rank 0 of world 1 / 2 / 4 / 8."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
for K in (int(a) for a in (sys.argv[1:] or ["16"])):
    for world in (1, 2, 4, 8):
        film = pt.Film(ctx, 1920, 1080)
        kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8, rank=0, world=world, pipeline=pt.PIPELINE_FUSED, frames_in_flight=K, sample_groups=1)
        p = pt.default_params(frame=0, frame_count=K, flags=pt.FLAG_PROFILE, **kw)
        pt.render(sc, film, p)
        rows = []
        for _ in range(5):
            film.clear(); ctx.reset_stats()
            t0 = time.perf_counter(); pt.render(sc, film, p); wall = (time.perf_counter() - t0) * 1e3
            st = ctx.stats()
            rows.append((wall, st.ms_total, st.ms_extend, st.rays))
        rows.sort()
        w, tot, ker, rays = rows[len(rows) // 2]
        print(f"K {K} world {world}: wall {w:.3f} ms, device {tot:.3f} ms, k_fused {ker:.3f} ms, rays {rays}, kernel rate {rays / ker / 1e3:.0f} Mrays/s")
        film.close()
