"""dev probe: per-rank efficiency of the pixel-tile sharding of BASELINE config C3, emulated on ONE GPU.  For world = 1, 2, 4, 8 every
rank's share of the 1920x1080 image is rendered alone (no collectives; K frames of 32 spp, default 32 = 1024 spp = C3) and timed:
    efficiency(N) = time(world 1) / (N x max over ranks of time(rank r of N))
-- what an N-GPU job would reach if the ranks ran side by side, the presentation gather (3.1 MB per rank, ~30 us) aside -- with the
per-rank ray counts (balance) and the AUTO shape each rank picked.  Writes one JSON object.
    python scripts/probe_shard_efficiency.py [K] [wavefront|fused|auto] > profiles/r04_shard_efficiency.json"""
import importlib, json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pipe = sys.argv[2] if len(sys.argv) > 2 else "wavefront"
pipeline = {"wavefront": pt.PIPELINE_WAVEFRONT, "fused": pt.PIPELINE_FUSED, "auto": pt.PIPELINE_AUTO}[pipe]
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
out = {"workload": f"C3: CornellBox-Original.obj 1920x1080, 32 spp/frame x {K} frames, 8 bounces, {pipe} pipeline; every rank of every world rendered alone on one MI355X",
       "worlds": {}}
base = None
for world in (1, 2, 4, 8):
    ranks = []
    for rank in range(world):
        film = pt.Film(ctx, 1920, 1080)
        kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8, rank=rank, world=world, pipeline=pipeline)
        p = pt.default_params(frame=0, frame_count=K, **kw)
        pt.render_prepare(sc, film, p)
        pt.render(sc, film, pt.default_params(frame=0, frame_count=2, **kw))       # warm-up
        ts = []
        for _ in range(3):
            film.clear(); ctx.reset_stats()
            t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        ranks.append({"rank": rank, "ms": round(statistics.median(ts) * 1e3, 3), "rays": st.rays, "frames_in_flight": st.frames_in_flight,
                      "sample_groups": st.sample_groups, "pipelines": st.pipelines, "workspace_bytes": st.workspace_bytes})
        film.close()
    slow = max(r["ms"] for r in ranks)
    if world == 1:
        base = slow
    rays = [r["rays"] for r in ranks]
    out["worlds"][str(world)] = {"ranks": ranks, "slowest_rank_ms": slow, "ideal_ms": round(base / world, 3),
                                 "efficiency": round(base / world / slow, 4), "aggregate_mrays_per_s": round(sum(rays) / (slow * 1e-3) / 1e6, 1),
                                 "rays_total": sum(rays), "rays_min_max": [min(rays), max(rays)],
                                 "ray_imbalance": round((max(rays) - min(rays)) / (sum(rays) / world), 5)}
    print(f"world {world}: slowest rank {slow:.2f} ms, ideal {base / world:.2f} ms, efficiency {100 * base / world / slow:.1f} %", file=sys.stderr)
print(json.dumps(out, indent=1))
