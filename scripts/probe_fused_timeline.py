"""dev probe (PT_LIB_AMD=build/variants/timeline/libpt_amd.so, built with -DPT_FUSED_TIMELINE): what the waves of ONE fused launch do in
time -- when each started, ran out of slots and ended (device clock, 100 MHz), and how many rays it traced.  Prints, per shape, the kernel's
span, when the FIRST wave found the slot counters empty, and the wave-time lost between a wave's end and the launch's end."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
L = pt.lib_amd()
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
NW = 256 * 8 * 8
buf = pt.DeviceBuffer(ctx, NW * 32)
L.pt_debug_fused_timeline.argtypes = [C.c_void_p]
assert L.pt_debug_fused_timeline(C.c_void_p(buf.ptr)) == 0
shapes = [(1, 1), (1, 32), (2, 1), (2, 16), (4, 1), (16, 1)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for K, G in shapes:
    film = pt.Film(ctx, 1920, 1080)
    p = pt.default_params(frame=0, frame_count=K, flags=pt.FLAG_PROFILE, width=1920, height=1080, spp_per_frame=32, max_depth=8,
                          pipeline=pt.PIPELINE_FUSED, frames_in_flight=K, sample_groups=G)
    pt.render(sc, film, p)
    film.clear(); ctx.reset_stats()
    pt.render(sc, film, p)
    st = ctx.stats()
    t = buf.read(np.uint64, (NW, 4))
    t = t[t[:, 2] > 0]
    t0 = t[:, 0].min()
    start, oos, end, rays = [(t[:, i].astype(np.int64) - int(t0)) / 100.0 for i in range(3)] + [t[:, 3].astype(np.int64)]   # us
    oos = np.where(t[:, 1] > 0, oos, end)
    span = end.max()
    q = lambda a, f: float(np.percentile(a, f))
    lost = float((span - end).sum() / (len(end) * span))
    print(f"K {K} G {G}: k_fused {st.ms_extend:.3f} ms, {len(end)} waves, span {span/1e3:.3f} ms | first out-of-slots at {oos.min()/1e3:.3f} ms, median {q(oos,50)/1e3:.3f}, last {oos.max()/1e3:.3f} | "
          f"wave ends: 1% {q(end,1)/1e3:.3f} 10% {q(end,10)/1e3:.3f} 50% {q(end,50)/1e3:.3f} 90% {q(end,90)/1e3:.3f} 99% {q(end,99)/1e3:.3f} max {span/1e3:.3f} | "
          f"wave-time after a wave's end {100*lost:.1f} % of waves x span | start spread {start.max():.1f} us | rays per wave min {rays.min()} median {int(np.median(rays))} max {rays.max()} | "
          f"{st.rays / st.ms_extend / 1e3:.0f} Mrays/s; if the launch ended at the median wave end: {st.rays / q(end,50) :.0f} Mrays/s")
    film.close()
