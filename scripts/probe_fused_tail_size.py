"""dev probe: samples per tail slot (pt_tuning.fused_tail_size) on the head + tail shapes; the films of every size must be the size-1 film."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H = 1920, 1080
for K, world in ((1, 1), (2, 1), (3, 1), (4, 1), (16, 8)):
    row, ref = [], None
    for rep in range(2):
        for size in (1, 2, 4):
            ctx.set_tuning(fused_tail_size=size)
            film = pt.Film(ctx, W, H)
            p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, rank=0, world=world)
            pt.render(sc, film, p)
            ts = []
            for _ in range(9):
                t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
            st = ctx.stats()
            film.clear(); ctx.reset_stats(); pt.render(sc, film, p)
            img = (film.read_f32().tobytes(), ctx.stats().rays)
            ref = ref or img
            row.append(f"size {size} (S {st.tail_samples}) {statistics.median(ts) * 1e3:.3f}{'' if img == ref else ' MISMATCH'}")
            film.close()
        row.append("|")
    print(f"K {K} world {world}: " + "  ".join(row), flush=True)
