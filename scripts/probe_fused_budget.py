"""dev check: the library default on C2's frame 0 under shrinking workspace budgets (pt_tuning.mem_budget_mb): the head + tail shape gives way to smaller shapes, the film stays the known answer."""
import importlib, sys, hashlib, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
ctx = pt.Context(0)
sc = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize_hashes.json")))["c2"]
W, H = gold["width"], gold["height"]
for budget in (-1, 4000, 1500, 400, 120, 60):
    ctx.set_tuning(mem_budget_mb=budget)
    film = pt.Film(ctx, W, H)
    try:
        ctx.reset_stats()
        pt.render(sc, film, pt.default_params(frame=0, frame_count=1, width=W, height=H, spp_per_frame=gold["spp_per_frame"], max_depth=gold["max_depth"]))
        st = ctx.stats()
        ok = hashlib.sha256(film.read_f32().astype("<f4").tobytes()).hexdigest() == gold["film_sha256"] and st.rays == gold["rays"]
        print(f"budget {budget} MB: pipeline {st.pipeline} groups {st.sample_groups} tail {st.tail_samples} workspace {st.workspace_bytes / 2**20:.0f} MB -> {'known answer' if ok else 'WRONG'}", flush=True)
    except pt.PtError as e:
        print(f"budget {budget} MB: error {e}", flush=True)
    film.close()
