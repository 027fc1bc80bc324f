"""Is the C2 speed level (23.1 / 25.0 / 25.7 / 26.7 Grays/s, probe_alloc_modes.py) tied to the allocation or to TIME?  One film, one
allocation; 16-frame renders back to back, an idle pause (no GPU work, no allocation) after every sixth."""
import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
v, i, f = pt.load_obj(pt.ASSET_CORNELL)
ctx = pt.Context(0)
scene = pt.Scene(ctx, v, i, f)
kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8)
film = pt.Film(ctx, 1920, 1080)
p = pt.default_params(frame=0, frame_count=16, **kw)
pt.render(scene, film, p)
pause = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
line = []
for rep in range(36):
    ctx.reset_stats()
    t0 = time.perf_counter()
    pt.render(scene, film, p)
    dt = time.perf_counter() - t0
    line.append(f"{ctx.stats().rays / dt / 1e6:.0f}")
    if rep % 6 == 5:
        line.append("|")
        time.sleep(pause)
print(f"pause {pause} s:", " ".join(line))
