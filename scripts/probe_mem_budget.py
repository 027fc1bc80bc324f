"""dev probe: what the workspace budget buys.  C4 (10 000 instances, 8 frames; wavefront and fused), C5 (1 M-triangle soup, 4 frames of 16 spp, depth 16), C5x (8 M, 2 frames)
through the wavefront pipeline's AUTO shapes under pt_tuning.mem_budget_mb = 2048 / 4096 / 8192 / 32768 / 0 (none): Mrays/s, the shape chosen, the workspace held.
usage: python scripts/probe_mem_budget.py [c4 c5 c5x]"""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
W, H = 1920, 1080
cfgs = sys.argv[1:] or ["c4", "c5", "c5x"]
SHAPES = {"c4": dict(spp=32, depth=8, frames=8), "c5": dict(spp=16, depth=16, frames=4, tris=1000000), "c5x": dict(spp=16, depth=16, frames=2, tris=8000000)}
for cfg in cfgs:
    sh = SHAPES[cfg]
    ctx = pt.Context(0)
    if cfg == "c4":
        sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
        sc.set_instances(pt.cornell_grid_instances())
    else:
        sc = pt.Scene(ctx, *pt.make_soup(sh["tris"], 1))
    for budget in (2048, 4096, 8192, 16384, 32768, 0):
        ctx.set_tuning(mem_budget_mb=budget)
        for pl_name, pl in (("wavefront", pt.PIPELINE_WAVEFRONT),) + ((("fused", pt.PIPELINE_FUSED),) if cfg == "c4" else ()):
            film = pt.Film(ctx, W, H)
            p = pt.default_params(frame=0, frame_count=sh["frames"], width=W, height=H, spp_per_frame=sh["spp"], max_depth=sh["depth"], pipeline=pl)
            try:
                pt.render(sc, film, p)
                ts = []
                for _ in range(3):
                    ctx.reset_stats()
                    t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
                st = ctx.stats()
                print(f"{cfg} budget {budget or 'none':>6} MB {pl_name:9s}: {st.rays / statistics.median(ts) / 1e6:9.1f} Mrays/s  {statistics.median(ts) * 1e3 / sh['frames']:8.3f} ms/frame  "
                      f"frames in flight {st.frames_in_flight} groups {st.sample_groups} pipelines {st.pipelines} workspace {st.workspace_bytes / 2**30:.2f} GB", flush=True)
            except pt.PtError as e:
                print(f"{cfg} budget {budget} MB {pl_name}: {e}", flush=True)
            film.close()
    sc.close(); ctx.close()
