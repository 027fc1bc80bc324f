"""dev probe: is the library's default shape (PT_PIPELINE_AUTO, nothing named) within 2 % of the best hand-picked shape at film sizes the rules were
not fitted on?  Cornell box, 32 spp, depth 8; per (size, frames per call): ms per call of the default, of the plain one-group shape, of all groups (32), of head + tail with
S = 4 .. 24, and of frames-in-flight halves where K > 1 -- and the best of them against the default.  Every shape's film is compared with the default's (bit-exact)."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
SIZES = [(1280, 720), (1024, 1024), (1920, 1080), (2560, 1440), (3840, 2160)]
if len(sys.argv) > 1:
    SIZES = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
worst = 0.0
for (W, H) in SIZES:
    for K in (1, 2, 20):
        def run(tune, **kw):
            old = ctx.set_tuning(**tune) if tune else {}
            try:
                film = pt.Film(ctx, W, H)
                p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, **kw)
                pt.render(sc, film, p)
                ts = []
                for _ in range(7 if K < 20 else 5):
                    film.clear(); ctx.reset_stats()
                    t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
                st = ctx.stats()
                img = film.read_f32().tobytes()
                film.close()
                return statistics.median(ts) * 1e3, st, img
            finally:
                if old:
                    ctx.set_tuning(**old)
        d_ms, d_st, d_img = run({})
        shapes = [("plain", dict(fused_tail=0), dict(sample_groups=1)), ("groups32", dict(fused_tail=0), dict(sample_groups=32))]
        shapes += [(f"tail{S}", dict(fused_tail=S), {}) for S in (4, 8, 10, 12, 16, 20, 24)]
        if K == 20:
            shapes += [("plain/fif10", dict(fused_tail=0), dict(sample_groups=1, frames_in_flight=10))]
        rows = []
        for name, tune, kw in shapes:
            try:
                ms, st, img = run(tune, pipeline=pt.PIPELINE_FUSED, **kw)
            except pt.PtError:      # (a hand-picked shape the library refuses: 32 groups x 20 frames of a big film exceed 2^31 slots)
                continue
            rows.append((ms, name, img == d_img and st.rays == d_st.rays))
        d2_ms, _, _ = run({})     # (the default once more, behind the others: the first shape of a film size pays for cold clocks and a fresh workspace)
        d_ms = min(d_ms, d2_ms)
        rows.sort()
        best_ms, best_name, _ = rows[0]
        gap = (d_ms / best_ms - 1.0) * 100.0
        worst = max(worst, gap)
        print(f"{W}x{H} K {K:2d}: default {d_ms:8.3f} ms (pipeline {d_st.pipeline}, groups {d_st.sample_groups}, tail {d_st.tail_samples}, frames in flight {d_st.frames_in_flight}, "
              f"{d_st.rays / d_ms / 1e6:.1f} Grays/s) | best {best_name} {best_ms:.3f} ms -> default is {gap:+.1f} % | "
              + "  ".join(f"{n} {m:.3f}{'' if ok else ' MISMATCH'}" for m, n, ok in rows[:6]), flush=True)
print(f"worst gap of the default against the best hand-picked shape: {worst:+.1f} %")
