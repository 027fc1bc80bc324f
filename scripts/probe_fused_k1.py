"""dev probe: one blocking fused pt_render per frame (the reference's dispatch shape, main.cpp:647-685), 24 frames, optionally with explicit sample
groups -- run under `rocprofv3 --kernel-trace --stats` to see what a frame's 6.5 ms are made of (k_fused<GROUPED>, k_resolve, memsets)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
film = pt.Film(ctx, 1920, 1080)
kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, frame_count=1, sample_groups=groups)
pt.render(sc, film, pt.default_params(frame=0, **kw))
ts = []
for k in range(1, 25):
    t0 = time.perf_counter(); pt.render(sc, film, pt.default_params(frame=k, **kw)); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
st = ctx.stats()
print(f"groups asked {groups} used {st.sample_groups}: median {ts[12]:.3f} ms, min {ts[0]:.3f}, workspace {st.workspace_bytes / 1e9:.2f} GB")
