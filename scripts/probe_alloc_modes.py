"""Is the C2 frame rate a property of the PROCESS or of the film's workspace allocation?  Creates and destroys the film (and with
it the 55 GB wavefront workspace) several times inside one process and times 16 frames on each; run it in a few processes."""
import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
v, i, f = pt.load_obj(pt.ASSET_CORNELL)
ctx = pt.Context(0)
scene = pt.Scene(ctx, v, i, f)
kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8)
out = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    film = pt.Film(ctx, 1920, 1080)
    p = pt.default_params(frame=0, frame_count=16, **kw)
    pt.render(scene, film, p)                       # allocates the workspace, warms up
    best = []
    for k in range(6):
        ctx.reset_stats()
        t0 = time.perf_counter()
        pt.render(scene, film, p)
        dt = time.perf_counter() - t0
        best.append(ctx.stats().rays / dt / 1e6)
    out.append((min(best), sorted(best)[3], max(best)))
    film.close()
print("pid", os.getpid(), "Mrays/s per allocation (min/median/max of 6 renders):", "  ".join(f"{a:.0f}/{b:.0f}/{c:.0f}" for a, b, c in out))
