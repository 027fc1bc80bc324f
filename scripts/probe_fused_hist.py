"""dev probe (PT_LIB_AMD=build/variants/hist/libpt_amd.so, built with -DPT_FUSED_HIST): rays started per 50 us over ONE fused launch at
1080p -- the machine's throughput in time, as a share of the launch's best 50 us, with the part started by waves on tail slots.
  python scripts/probe_fused_hist.py K:S[:G] ...     (S: pt_tuning.fused_tail, -1 = the rule; G: explicit sample groups, with S = 0)"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
L = pt.lib_amd()
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
W, H = 1920, 1080
buf = pt.DeviceBuffer(ctx, 16384 * 16 * 4)
L.pt_debug_fused_hist.argtypes = [C.c_void_p]
assert L.pt_debug_fused_hist(C.c_void_p(buf.ptr)) == 0
for a in sys.argv[1:] or ["1:-1", "1:0:1", "1:0:32", "2:-1", "2:0:1"]:
    v = [int(x) for x in a.split(":")]
    K, S, G = v[0], v[1], (v[2] if len(v) > 2 else 0)
    ctx.set_tuning(fused_tail=S)
    film = pt.Film(ctx, W, H)
    p = pt.default_params(frame=0, frame_count=K, flags=pt.FLAG_PROFILE, width=W, height=H, spp_per_frame=32, max_depth=8, pipeline=pt.PIPELINE_FUSED, sample_groups=G)
    pt.render(sc, film, p)
    film.clear(); ctx.reset_stats()
    buf.write(np.zeros(16384 * 16, np.uint32))
    pt.render(sc, film, p)
    st = ctx.stats()
    h = buf.read(np.uint32, (2, 8192, 16)).astype(np.int64).sum(axis=2)
    nz = np.nonzero(h[0])[0]
    # (the 8192 buckets wrap every 0.41 s: rotate so that the launch is contiguous)
    gaps = np.diff(np.concatenate([nz, [nz[0] + 8192]]))
    first = nz[(int(np.argmax(gaps)) + 1) % len(nz)]
    idx = (first + np.arange(8192)) % 8192
    all_, tail = h[0][idx], h[1][idx]
    n = int(np.nonzero(all_)[0].max()) + 1
    all_, tail = all_[:n], tail[:n]
    peak = np.sort(all_)[-max(3, n // 10):].mean()      # the mean of the best tenth of the buckets
    print(f"K {K} S {S} G {G} -> tail {st.tail_samples} groups {st.sample_groups}: k_fused {st.ms_extend:.3f} ms, {n} buckets of 50 us, rays {int(all_.sum())}; "
          f"best-tenth rate {peak / 50e-6 / 1e9:.1f} Grays/s; at that rate the launch's rays take {all_.sum() / peak * 0.05:.3f} ms")
    step = max(1, n // 32)
    print(f"   percent of best-tenth rate per {50 * step} us: " + " ".join(f"{100 * all_[i:i + step].mean() / peak:.0f}" for i in range(0, n, step)))
    print("   of which on tail slots:                 " + " ".join(f"{100 * tail[i:i + step].mean() / peak:.0f}" for i in range(0, n, step)))
    print(f"   absolute Grays/s per {50 * step} us: " + " ".join(f"{all_[i:i + step].mean() / 50e-6 / 1e9:.1f}" for i in range(0, n, step)))
    print("   Mrays per 50 us, first 24 buckets: " + " ".join(f"{x / 1e6:.2f}" for x in all_[:24]))
    print("   Mrays per 50 us, last 40 buckets:  " + " ".join(f"{x / 1e6:.2f}" for x in all_[-40:]))
    film.close()
