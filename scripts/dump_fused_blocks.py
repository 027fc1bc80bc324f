"""Where the fused kernel's VALU instructions go: wave executions and lanes of every block of k_fused's loop (the instrumented twin,
PT_FLAG_COUNT_VISITS on PT_PIPELINE_FUSED) x the blocks' VALU instruction counts in the shipped ISA (profiles/isa_valu_model.json "k_fused",
written by scripts/isa_regions.py --json).
usage: python scripts/dump_fused_blocks.py [frames=16] [W=1920] [H=1080] [refill=...]"""
import importlib, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")


def fused_block_table(pt, ctx, scene, W, H, frames, spp=32, depth=8, model=None):
    """-> dict: per block waves / lanes / valu, totals per 64 walked rays, lanes per VALU instruction"""
    model = model or json.load(open(os.path.join(REPO, "profiles", "isa_valu_model.json")))["k_fused"]
    film = pt.Film(ctx, W, H)
    kw = dict(width=W, height=H, spp_per_frame=spp, max_depth=depth)
    old = ctx.set_tuning(cull=1)
    try:
        ctx.reset_stats()
        pt.render(scene, film, pt.default_params(frame=0, frame_count=frames, pipeline=pt.PIPELINE_FUSED, flags=pt.FLAG_COUNT_VISITS, **kw))
    finally:
        ctx.set_tuning(**old)
    st = ctx.stats()
    bc = ctx.block_counts()
    film.close()
    walked = st.rays - st.rays_culled
    rows, tot, lanes = {}, 0.0, 0.0
    for name, (waves, ln) in bc.items():
        v = model["blocks"][name]["valu"]
        tot += waves * v
        lanes += ln * v
        rows[name] = {"waves_per_64_rays": round(waves / walked * 64.0, 3), "lanes": round(ln / waves, 1) if waves else None, "valu": v,
                      "valu_per_64_rays": round(waves * v / walked * 64.0, 1)}
    return {"rays": st.rays, "walked_rays": walked, "tail_samples": st.tail_samples, "sample_groups": st.sample_groups,
            "valu_wave_instr_per_64_rays": round(tot / walked * 64.0, 1), "valu_active_lanes_per_instr": round(lanes / max(tot, 1.0), 1),
            "isa_revision": model.get("revision"), "blocks": rows}


if __name__ == "__main__":
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    v, i, f = pt.load_obj(pt.ASSET_CORNELL)
    ctx = pt.Context(0)
    for a in sys.argv[4:]:
        k, val = a.split("=")
        ctx.set_tuning(**{k: int(val)})
    scene = pt.Scene(ctx, v, i, f)
    t = fused_block_table(pt, ctx, scene, W, H, frames)
    print(f"{W}x{H} x {frames} frames: rays {t['rays']} walked {t['walked_rays']} tail {t['tail_samples']} groups {t['sample_groups']}")
    print(f"{'block':9s} {'waves/64 rays':>14s} {'lanes':>6s} {'VALU':>5s} {'VALU/64 rays':>13s} {'share':>6s} {'lane-instr/ray':>15s}")
    for n, r in t["blocks"].items():
        li = r["valu_per_64_rays"] / 64.0 * (r["lanes"] or 0)
        print(f"{n:9s} {r['waves_per_64_rays']:14.3f} {str(r['lanes']):>6s} {r['valu']:5.0f} {r['valu_per_64_rays']:13.1f} "
              f"{100.0 * r['valu_per_64_rays'] / t['valu_wave_instr_per_64_rays']:5.1f}% {li:15.1f}")
    print(f"model: {t['valu_wave_instr_per_64_rays']} VALU wave-instructions per 64 walked rays at {t['valu_active_lanes_per_instr']} lanes "
          f"(PMC of round 5: 1508 at 34.4)")
    print(json.dumps(t))
