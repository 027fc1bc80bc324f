"""dev probe: config C4 (10 000 instances) through the fused two-level kernel: ms per frame by frames per call and sample groups, cull off / on."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
sc.set_instances(pt.cornell_grid_instances())
W, H = 1920, 1080
ref = {}
for K in (1, 2, 4, 8, 16):
    for cull in (0, 1):
        row = []
        for G in (0, 1, 2, 4, 8, 32):
            ctx.set_tuning(cull=cull)
            film = pt.Film(ctx, W, H)
            p = pt.default_params(frame=0, frame_count=K, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, sample_groups=G)
            pt.render(sc, film, p)
            ts = []
            for _ in range(3 if K >= 8 else 5):
                film.clear(); ctx.reset_stats()
                t0 = time.perf_counter(); pt.render(sc, film, p); ts.append(time.perf_counter() - t0)
            st = ctx.stats()
            img = film.read_f32().tobytes()
            ref.setdefault(K, (img, st.rays))
            ok = img == ref[K][0] and st.rays == ref[K][1]
            row.append(f"G{G}{'->' + str(st.sample_groups) if G == 0 else ''} {statistics.median(ts) * 1e3 / K:.3f}{'' if ok else ' MISMATCH'}")
            film.close()
        print(f"C4 K {K} cull {cull} (rays per frame {st.rays / K / 1e6:.1f} M, culled {st.rays_culled / K / 1e6:.1f} M): ms per FRAME: " + "  ".join(row), flush=True)
