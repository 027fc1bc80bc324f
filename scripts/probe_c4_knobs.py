"""dev probe: the fused two-level kernel's waiting rules on config C4 with the cull (16 frames per call): one knob at a time from the defaults
(refill 48, enter_min 16, leaf_min 8, node_yield 6)."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
sc.set_instances(pt.cornell_grid_instances())
W, H, K = 1920, 1080, 16
film = pt.Film(ctx, W, H)
p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED)
def run(**knobs):
    old = ctx.set_tuning(**knobs)
    try:
        pt.render(sc, film, p)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        return statistics.median(ts) * 1e3 / K
    finally:
        ctx.set_tuning(**old)
print(f"defaults: {run():.3f} ms per frame", flush=True)
for name, vals in (("refill", (32, 40, 44, 52, 56, 60)), ("enter_min", (4, 8, 24, 32, 48)), ("leaf_min", (1, 4, 16, 24, 32)), ("node_yield", (0, 2, 3, 4, 8, 12))):
    print(name + ": " + "  ".join(f"{v}: {run(**{name: v}):.3f}" for v in vals), flush=True)
print(f"defaults again: {run():.3f}", flush=True)
