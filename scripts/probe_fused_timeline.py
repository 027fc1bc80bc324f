"""dev probe (PT_LIB_AMD=build/variants/timeline/libpt_amd.so, built with -DPT_FUSED_TIMELINE): what the waves of ONE fused launch do in
time -- when each started, ran out of slots and ended (device clock, 100 MHz), how many rays it traced; and per LANE the last slot it
completed with its start and end.  Prints, per shape, the kernel's span, when the FIRST wave found the slot counters empty, the wave-time
lost between a wave's end and the launch's end, and what the slots that end late are (pixel ring, frame, duration)."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
L = pt.lib_amd()
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H = 1920, 1080
NW = 256 * 3 * 8
NL = NW * 64
buf = pt.DeviceBuffer(ctx, NW * 32 + NL * 24)
L.pt_debug_fused_timeline.argtypes = [C.c_void_p]
assert L.pt_debug_fused_timeline(C.c_void_p(buf.ptr)) == 0
# the tile order of the fused pipeline with one group (film_work.hip): centre first by Chebyshev ring
tx_n, ty_n = (W + 7) // 8, (H + 7) // 8
tiles = [(tx, ty) for ty in range(ty_n) for tx in range(tx_n)]
ring = lambda t: max(abs(t[0] + 0.5 - tx_n / 2) / (tx_n / 2), abs(t[1] + 0.5 - ty_n / 2) / (ty_n / 2))
tiles_sorted = sorted(tiles, key=ring)          # (python's sort is stable like std::stable_sort)
ring_of_tile = np.array([ring(t) for t in tiles_sorted])
shapes = [(1, 1), (2, 1), (16, 1)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for K, G in shapes:
    film = pt.Film(ctx, W, H)
    p = pt.default_params(frame=0, frame_count=K, flags=pt.FLAG_PROFILE, width=W, height=H, spp_per_frame=32, max_depth=8,
                          pipeline=pt.PIPELINE_FUSED, frames_in_flight=K, sample_groups=G)
    pt.render(sc, film, p)
    film.clear(); ctx.reset_stats()
    pt.render(sc, film, p)
    st = ctx.stats()
    raw = buf.read(np.uint64, (NW * 4 + NL * 3,))
    t = raw[:NW * 4].reshape(NW, 4)
    lanes = raw[NW * 4:].reshape(NL, 3)
    t = t[t[:, 2] > 0]
    t0 = t[:, 0].min()
    start, oos, end, rays = [(t[:, i].astype(np.int64) - int(t0)) / 100.0 for i in range(3)] + [(t[:, 3] & np.uint64((1 << 48) - 1)).astype(np.int64)]   # us
    items = (t[:, 3] >> np.uint64(48)).astype(np.int64)   # donated items the wave created (fused_kernel.h part 2b)
    oos = np.where(t[:, 1] > 0, oos, end)
    span = end.max()
    q = lambda a, f: float(np.percentile(a, f))
    lost = float((span - end).sum() / (len(end) * span))
    print(f"K {K} G {G}: k_fused {st.ms_extend:.3f} ms, {len(end)} waves, span {span/1e3:.3f} ms | first out-of-slots at {oos.min()/1e3:.3f} ms, median {q(oos,50)/1e3:.3f}, last {oos.max()/1e3:.3f} | "
          f"wave ends: 1% {q(end,1)/1e3:.3f} 10% {q(end,10)/1e3:.3f} 50% {q(end,50)/1e3:.3f} 90% {q(end,90)/1e3:.3f} 99% {q(end,99)/1e3:.3f} max {span/1e3:.3f} | "
          f"wave-time after a wave's end {100*lost:.1f} % of waves x span | rays per wave min {rays.min()} median {int(np.median(rays))} max {rays.max()} | "
          f"{st.rays / st.ms_extend / 1e3:.0f} Mrays/s; if the launch ended at the median wave end: {st.rays / q(end,50) :.0f} Mrays/s | "
          f"donated items: {int(items.sum())} in all, per wave median {int(np.median(items))} max {int(items.max())}, waves with none {int((items == 0).sum())}")
    if G == 1:
        ok = (lanes[:, 0] & np.uint64(0xFFFFFFFF)) != np.uint64(0xFFFFFFFF)
        ls = lanes[ok]
        slot = (ls[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        nslots = (ls[:, 0] >> np.uint64(32)).astype(np.int64)
        s0 = (ls[:, 1].astype(np.int64) - int(t0)) / 100.0
        s1 = (ls[:, 2].astype(np.int64) - int(t0)) / 100.0
        spl = len(tiles) * 64
        local = slot % spl
        rg = ring_of_tile[local // 64]
        dur = s1 - s0
        print(f"   lanes' LAST slots: {len(ls)} lanes, slots per lane median {int(np.median(nslots))}; duration us: median {q(dur,50):.0f} 90% {q(dur,90):.0f} 99% {q(dur,99):.0f} max {dur.max():.0f}")
        for lo, hi in ((0, 50), (50, 90), (90, 99), (99, 100)):
            a, b = q(s1, lo), q(s1, hi) if hi < 100 else s1.max() + 1
            m = (s1 >= a) & (s1 < b)
            if m.sum() == 0:
                continue
            print(f"   last slots ending in the {lo}-{hi}% window [{a/1e3:.3f}, {b/1e3:.3f}) ms: n {int(m.sum())}, interior (ring < 0.75) {100*float((rg[m] < 0.75).mean()):.0f} %, ring median {q(rg[m],50):.2f}, "
                  f"began median {q(s0[m],50)/1e3:.3f} ms (min {s0[m].min()/1e3:.3f}), duration median {q(dur[m],50):.0f} us max {dur[m].max():.0f}")
    film.close()
