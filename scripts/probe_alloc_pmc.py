"""dev probe: which counter separates a slow film allocation from a fast one?
Step 1 (this file as a program, inside rocprofv3): one process allocates the C2 film N times (the 133 M-slot shape of round 3 --
16 frames x 4 groups, 51 GB --, where the levels 23 / 25 / 26-27 Grays/s were frequent) and renders R times on each; prints one line
per allocation.  Step 2 (`analyse <counter_collection.csv> <R>`): cuts the dispatch stream into allocations at every R-th k_resolve
and sums, per allocation, every counter over the k_shade (and k_extend) dispatches."""
import csv, importlib, json, os, sys, time
from collections import defaultdict


def run(n_alloc, renders):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
    ctx = pt.Context(0)
    scene = pt.Scene(ctx, *pt.load_obj(pt.ASSET_CORNELL))
    kw = dict(width=1920, height=1080, spp_per_frame=32, max_depth=8, frame=0, frame_count=16, frames_in_flight=16,
              sample_groups=int(os.environ.get("PROBE_GROUPS", "4")))
    for a in range(n_alloc):
        film = pt.Film(ctx, 1920, 1080)
        vals = []
        for r in range(renders):
            film.clear()
            ctx.reset_stats()
            t0 = time.perf_counter()
            pt.render(scene, film, pt.default_params(**kw))
            vals.append(ctx.stats().rays / (time.perf_counter() - t0) / 1e6)
        print(f"alloc {a}: Mrays/s " + " ".join(f"{v:.0f}" for v in vals) + f"  workspace GB {ctx.stats().workspace_bytes / 2**30:.1f}", flush=True)
        film.close()


def analyse(path, renders):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    alloc, resolves, seen = 0, 0, None
    acc = defaultdict(lambda: defaultdict(float))
    for r in rows:
        k = r["Kernel_Name"]
        short = "shade" if "k_shade" in k else "extend" if "k_extend" in k else "resolve" if "k_resolve" in k else None
        if short == "resolve" and r["Dispatch_Id"] != seen:
            seen = r["Dispatch_Id"]
            resolves += 1
            if resolves % renders == 0:
                alloc += 1
            continue
        if short in ("shade", "extend"):
            acc[(alloc, short)][r["Counter_Name"]] += float(r["Counter_Value"])
    out = {}
    for (a, s), c in sorted(acc.items()):
        out.setdefault(str(a), {})[s] = {k: v for k, v in c.items()}
    return out


if __name__ == "__main__":
    if sys.argv[1] == "analyse":
        print(json.dumps(analyse(sys.argv[2], int(sys.argv[3])), indent=0))
    else:
        run(int(sys.argv[1]), int(sys.argv[2]))
