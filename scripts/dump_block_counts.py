"""Wave-level block counts of the traversal kernel of a config (PT_FLAG_COUNT_VISITS), per 64 rays, with the lanes inside.
usage: python scripts/dump_block_counts.py c2|c4 [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
v, i, f = pt.load_obj(pt.ASSET_CORNELL)
ctx = pt.Context(0)
scene = pt.Scene(ctx, v, i, f)
if cfg == "c4":
    scene.set_instances(pt.cornell_grid_instances())
film = pt.Film(ctx, 1920, 1080)
kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8)
pt.render(scene, film, pt.default_params(frame=0, frame_count=frames, **kw))
ctx.reset_stats()
pt.render(scene, film, pt.default_params(frame=0, frame_count=frames, flags=pt.FLAG_COUNT_VISITS, **kw))
st = ctx.stats()
u = st.rays / 64.0
print(f"{cfg}: rays {st.rays}  nodes/ray {st.nodes_visited / st.rays:.2f}  tris/ray {st.tris_tested / st.rays:.2f}")
rows = [("iterations", st.wave_iterations, 64 * st.wave_iterations), ("refills", st.wave_refills, st.rays),
        ("node steps", st.node_steps, st.nodes_visited), ("pop iterations", st.wave_pops, st.pop_lanes),
        ("leaf steps", st.tri_steps, st.leaf_lanes), ("hit blocks", st.wave_hit_blocks, st.hit_lanes),
        ("instance entries", st.enter_steps, st.enter_lanes), ("finishes", st.wave_finishes, st.rays)]
for name, waves, lanes in rows:
    if waves:
        print(f"  {name:18s} wave executions per 64 rays {waves / u:7.3f}   lanes inside {lanes / waves:5.1f}   lane executions per ray {lanes / st.rays:6.3f}")
