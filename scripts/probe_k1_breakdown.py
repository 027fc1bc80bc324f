"""dev probe: K = 1 / K = 2 blocking fused calls at 1080p under rocprofv3 --kernel-trace: kernel time against wall time per call."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
W, H = 1920, 1080
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else -1
ctx.set_tuning(fused_tail=S)
film = pt.Film(ctx, W, H)
p = pt.library_default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8) if hasattr(pt, "library_default_params") else \
    pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8)
pt.render(sc, film, p)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
st = ctx.stats()
print(f"K {K} S {S} -> tail {st.tail_samples} groups {st.sample_groups}: wall per call median {statistics.median(ts) * 1e3:.3f} ms  min {min(ts) * 1e3:.3f}", flush=True)
