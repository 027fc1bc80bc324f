"""dev probe: the 8-wide kernel's leaf-phase waiting rules re-swept on C5 (1 M-triangle soup, 4 frames of 16 spp, depth 16) and C5x (8 M, 2 frames),
workspace budget 32 GB: tri_enter (lanes that wait with leaf triangles before a triangle step runs against a majority of descending lanes) x tri_stay (a triangle step
repeats while at least this many lanes still hold one; 65 = never).  usage: python scripts/probe_c5_knobs.py [c5|c5x]"""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
tris, frames = (1000000, 4) if cfg == "c5" else (8000000, 2)
ctx = pt.Context(0)
ctx.set_tuning(mem_budget_mb=32768)
sc = pt.Scene(ctx, *pt.make_soup(tris, 1))
W, H = 1920, 1080
film = pt.Film(ctx, W, H)
p = pt.default_params(frame=0, frame_count=frames, width=W, height=H, spp_per_frame=16, max_depth=16, pipeline=pt.PIPELINE_WAVEFRONT)
pt.render(sc, film, p)
ref = film.read_f32().tobytes()
for te in (-1, 4, 8, 12, 16, 24, 32):
    row = []
    for ts_ in (-1, 8, 16, 24, 32, 65):
        old = ctx.set_tuning(tri_enter=te, tri_stay=ts_)
        ts = []
        for _ in range(3):
            film.clear(); ctx.reset_stats()
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        ok = film.read_f32().tobytes() == ref
        ctx.set_tuning(**old)
        row.append(f"stay {ts_:3d}: {st.rays / statistics.median(ts) / 1e6:7.1f}{'' if ok else ' MISMATCH'}")
    print(f"{cfg} tri_enter {te:3d} | " + " | ".join(row), flush=True)
