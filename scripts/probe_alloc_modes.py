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
ballast_gb = float(os.environ.get('PROBE_BALLAST_GB', '0'))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ballast = [pt.DeviceBuffer(ctx, int(8 * 2**30)) for _ in range(int(ballast_gb / 8))] if os.environ.get('PROBE_BALLAST_FIRST') else []
    film = pt.Film(ctx, 1920, 1080)
    extra = dict(sample_groups=int(sys.argv[2]), frames_in_flight=int(sys.argv[3])) if len(sys.argv) > 3 else {}
    p = pt.default_params(frame=0, frame_count=int(sys.argv[4]) if len(sys.argv) > 4 else 16, **extra, **kw)
    if not os.environ.get('PROBE_BALLAST_FIRST'):
        ballast = [pt.DeviceBuffer(ctx, int(8 * 2**30)) for _ in range(int(ballast_gb / 8))]
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
    for b in ballast:
        b.close()
st = ctx.stats()
print("workspace GB %.1f redone %d groups %d fif %d |" % (st.workspace_bytes / 2**30, st.redone_batches, st.sample_groups, st.frames_in_flight), "pid", os.getpid(), "Mrays/s per allocation (min/median/max of 6 renders):", "  ".join(f"{a:.0f}/{b:.0f}/{c:.0f}" for a, b, c in out))
