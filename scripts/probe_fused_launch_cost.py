"""dev probe: what a fused LAUNCH costs beyond its frames: 32 frames at 1080p in one call as 1, 2, 4, 8, 16, 32 launches (frames_in_flight = 32 .. 1;
the launches of a call follow each other on the stream without a host wait), plain one-group shape; then the same call after an idle gap."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H, spp = 1920, 1080, 32
film = pt.Film(ctx, W, H)
for fif in (32, 16, 8, 4, 2, 1):
    p = pt.default_params(frame=0, frame_count=32, width=W, height=H, spp_per_frame=spp, max_depth=8, pipeline=pt.PIPELINE_FUSED, sample_groups=1, frames_in_flight=fif)
    pt.render(sc, film, p)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
    m = statistics.median(ts) * 1e3
    print(f"32 frames as {32 // fif} launches of {fif}: {m:.2f} ms per call -> {(m - 5.32 * 32) / (32 // fif):.3f} ms per launch over 5.32 ms per frame", flush=True)
for K in (1, 4):
    p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=spp, max_depth=8, pipeline=pt.PIPELINE_FUSED, sample_groups=1)
    pt.render(sc, film, p)
    for gap in (0.0, 0.001, 0.02, 0.2):
        ts = []
        for _ in range(9):
            time.sleep(gap)
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        print(f"K {K} after {gap * 1e3:.0f} ms of idle: {statistics.median(ts) * 1e3:.3f} ms per call", flush=True)
