"""dev tool: repeated renders of the same work must give bit-identical films (race detector for the
two-pipeline scheduling, lane refill, compaction atomics and term logs)."""
import hashlib, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0); sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL)); film = pt.Film(ctx, 1920, 1080)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cases = [dict(frame_count=8), dict(frame_count=8, sample_groups=4, frames_in_flight=2), dict(frame_count=3, rank=1, world=3),
         dict(frame_count=5, frames_in_flight=1, sample_groups=1),
         # the fused single-kernel pipeline: dynamic slot hand-out, work stealing between the eight counters, term logs
         dict(frame_count=8, pipeline=pt.PIPELINE_FUSED), dict(frame_count=2, pipeline=pt.PIPELINE_FUSED),
         dict(frame_count=3, rank=1, world=3, pipeline=pt.PIPELINE_FUSED), dict(frame_count=1, sample_groups=32, pipeline=pt.PIPELINE_FUSED),
         # the library default (head + tail slots at one and two frames per call, the cull), and an off-centre view
         dict(frame_count=1, pipeline=pt.PIPELINE_AUTO), dict(frame_count=2, pipeline=pt.PIPELINE_AUTO),
         dict(frame_count=4, pipeline=pt.PIPELINE_AUTO, cam_origin=(1.0, -0.2, 5.0), cam_target=(1.0, -0.2, 2.0))]
t0 = time.time()
for c in cases:
    hashes = set(); rays = set()
    for i in range(N):
        film.clear(); ctx.reset_stats()
        pt.render(sc, film, pt.default_params(width=1920, height=1080, spp_per_frame=32, max_depth=8, **c))
        hashes.add(hashlib.sha256(film.read_f32().tobytes()).hexdigest()); rays.add(ctx.stats().rays)
    print(c, "distinct films:", len(hashes), "distinct ray counts:", len(rays), "OK" if len(hashes) == 1 and len(rays) == 1 else "MISMATCH")
    assert len(hashes) == 1 and len(rays) == 1
print("soak ok in %.1f s" % (time.time() - t0))
