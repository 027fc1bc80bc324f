#!/usr/bin/env python3
"""bench.py -- the headline benchmark: Mrays/s (and ms/frame) of the radiance loop on the Cornell
box at 1920x1080, 8 bounces (BASELINE.json metric), on N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one frame = one reference launch (traceRaysKHR(W,H,1), main.cpp:659): 32 samples per
pixel, <= 8 rays each, blended into the film (raygen.rgen:41-91).  K steps = 32*K spp; K=2 is
exactly config C2 (64 spp); the default K=16 (512 spp) lets the device keep 16 frames in flight.  The scene + LBVH are resident in HBM before the timed
region; the timed region is the K frames (all kernels: generate, extend, shade/compact, resolve)
plus, for N>1, the one RCCL reduce of the float film to rank 0.  Rank 0 prints ONE JSON line.

N>1 shards the 8x8 pixel tiles of the SAME image over the ranks (config C3's decomposition), so
the total work per step is fixed: "scaling": "strong".
"""
import argparse
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
# algorithmic bytes per ray of the wavefront pipeline (SURVEY.md section 8d / DESIGN.md section 7)
BYTES_EXTEND = 40.0    # read ray 28 (index-free dense queue: 24 + 4 slot id passed along), write hit 12
BYTES_SHADE = 104.0
BYTES_PER_PATH = 96.0


def cpu_baseline(arrays, name, width, height, spp_max, depth, budget_s=10.0, instances=None):
    """The oracle (a CPU port of the reference shaders + software LBVH) timed on this host, on a
    bounded sample of the same workload: the same image, all cores, as many samples per pixel as fit
    into ~budget_s seconds (calibrated with a 1-spp pass, at most spp_max)."""
    from oracle import pt_oracle as orc
    osc = orc.Scene(*arrays)
    if instances is not None:
        osc.set_instances(instances)      # two-level scene (config C4): the oracle walks its own TLAS
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    osc.render_frame(orc.default_params(width=width, height=height, spp_per_frame=1, max_depth=depth), mode=1, nthreads=cores)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(spp_max, budget_s / max(t1, 1e-3))))
    p = orc.default_params(width=width, height=height, spp_per_frame=spp, max_depth=depth)
    t0 = time.perf_counter()
    img, rays, cnt, _ = osc.render_frame(p, mode=1, nthreads=cores)
    dt = time.perf_counter() - t0
    base = {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port", "_rays": rays, "_spp": spp, "_img": img,
            "sample": f"{name} {width}x{height}, {spp} spp (frame 0), depth {depth}: {rays} rays in "
                      f"{dt:.2f} s; oracle/pt_oracle.c, software LBVH, gcc -O2 -ffp-contract=off, {cores} threads"}
    return base, cnt.nodes_visited / max(rays, 1), cnt.tris_tested / max(rays, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--config", choices=["c2", "c4", "c5"], default="c2",
                    help="c2 = Cornell box (the headline), c4 = Cornell x 10 000 instances (two-level BVH), "
                         "c5 = 1M-triangle soup, 16 spp/frame, depth 16")
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--soup-tris", type=int, default=1000000)
    ap.add_argument("--frames-in-flight", type=int, default=0)
    ap.add_argument("--sample-groups", type=int, default=0)
    ap.add_argument("--extend", choices=["auto", "flat", "lds", "hbm"], default="auto", help="closest-hit kernel variant")
    ap.add_argument("--bvh-quality", choices=["fast_trace", "fast_build"], default="fast_trace",
                    help="fast_trace = the reference's ePreferFastTrace (main.cpp:419, default); fast_build = collapsed LBVH only")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not hipEvent-time each extend/shade launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=10.0,
                    help="seconds of oracle time for cpu_baseline (it renders as many spp of frame 0 as fit; when that is the "
                         "whole frame, film and ray count are compared with the GPU's)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU fallback)")
    # PT_BENCH_EMULATE=1 (dev check of the N > 1 code path on a 1-GPU box): every rank uses GPU 0 and the
    # collectives run over gloo on host copies.  Never used for reported numbers.
    emulate = os.environ.get("PT_BENCH_EMULATE") == "1" and world > 1
    if emulate:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if emulate else dev     # where the collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emulate:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def reduce_film(t):
        if emulate:
            h = t.cpu()
            ptd.reduce_film(h, dst=0)
            if rank == 0:
                t.copy_(h)
        else:
            ptd.reduce_film(t, dst=0)

    pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
    ptd = importlib.import_module("single-file-vulkan-pathtracing_amd.distributed")

    W, H = args.width, args.height
    if args.config == "c5":
        args.spp = args.spp or 16
        args.depth = args.depth or 16
        obj = f"/tmp/pt_soup_{args.soup_tris}_rank{rank}.obj"     # generated, not committed (139 MB of text)
        t0 = time.perf_counter()
        pt.write_soup_obj(obj, args.soup_tris, 1)
        t1 = time.perf_counter()
        arrays = pt.load_obj(obj)
        ingest = {"generate_s": round(t1 - t0, 3), "load_obj_s": round(time.perf_counter() - t1, 3),
                  "obj_bytes": os.path.getsize(obj)}
        os.remove(obj)
        scene_name = f"soup {args.soup_tris} triangles (PCG seed 1)"
    else:
        args.spp = args.spp or 32
        args.depth = args.depth or 8
        arrays = pt.load_obj(pt.ASSET_CORNELL)
        ingest = None
        scene_name = "CornellBox-Original.obj"
    stream = torch.cuda.current_stream(dev)
    ctx = pt.Context(local_rank, stream=stream.cuda_stream)
    scene = pt.Scene(ctx, *arrays)          # upload + on-device LBVH build (untimed, reported apart)
    if args.bvh_quality == "fast_build":
        scene.set_bvh_quality(pt.BVH_PREFER_FAST_BUILD)
    if args.config == "c4":
        t0 = time.perf_counter()
        scene.set_instances(pt.cornell_grid_instances())      # TLAS build on device
        ctx.sync()
        tlas_ms = (time.perf_counter() - t0) * 1e3
        scene_name = "CornellBox-Original.obj x 10 000 instances (100x100 grid, scale 0.009)"
    info = scene.info()
    film_t = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)   # torch owns the film: RCCL reduces it in place
    film = pt.Film(ctx, W, H, device_ptr=film_t.data_ptr())
    flags = 0 if args.no_kernel_events else pt.FLAG_PROFILE
    common = dict(width=W, height=H, spp_per_frame=args.spp, max_depth=args.depth, rank=rank, world=world,
                  frames_in_flight=args.frames_in_flight, sample_groups=args.sample_groups,
                  extend={"auto": pt.EXTEND_AUTO, "flat": pt.EXTEND_FLAT, "lds": pt.EXTEND_LDS, "hbm": pt.EXTEND_HBM}[args.extend])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # the workspace is sized once for the timed call's shape (frames in flight x sample groups) ...
    timed = pt.default_params(frame=0, frame_count=args.steps, flags=flags, **common)
    pt.render_prepare(scene, film, timed)
    shape = ctx.stats()
    common.update(frames_in_flight=shape.frames_in_flight, sample_groups=shape.sample_groups)
    timed = pt.default_params(frame=0, frame_count=args.steps, flags=flags, **common)
    # ... then W untimed warm-up frames run through the same kernels
    if args.warmup > 0:
        pt.render(scene, film, pt.default_params(frame=0, frame_count=args.warmup, **common))
    if world > 1:      # communicator set-up (the first RCCL collective of a process) is not a step: always outside the timed region
        tmp = film_t.clone()
        reduce_film(tmp)
        del tmp
    film.clear()
    ctx.reset_stats()

    barrier()
    t0 = time.perf_counter()
    pt.render(scene, film, timed)
    reduce_film(film_t)                 # the one collective per presented image (no-op for N=1)
    barrier()
    dt = time.perf_counter() - t0

    st = ctx.stats()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    rays_total, paths_total = ptd.sum_counters([st.rays, st.paths], cdev)
    rays_minmax = None
    if world > 1:
        lo = torch.tensor([st.rays], dtype=torch.int64, device=cdev)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        rays_minmax = [int(lo.item()), int(hi.item())]

    if rank == 0:
        mean_len = rays_total / max(paths_total, 1)
        out = {
            "metric": {"c2": "Mrays/s, Cornell Box 1920x1080 @ 8 bounces (ms/frame in ms_per_step)",
                       "c4": "Mrays/s, Cornell Box x 10k instances (two-level BVH) 1920x1080 @ 8 bounces (BASELINE config C4)",
                       "c5": "Mrays/s, 1M-triangle soup 1920x1080 @ 16 bounces (BASELINE config C5)"}[args.config],
            "value": round(rays_total / dt / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 4),
            "ms_per_1spp_pass": round(dt * 1e3 / args.steps / args.spp, 5),   # also per sample-per-pixel pass
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("CornellBox-Original.obj (the reference's own scene, 36 triangles)" if args.config in ("c2", "c4") else
                     "synthetic triangle soup (generator pth_write_soup_obj, seed 1), written as OBJ+MTL and parsed by the host loader")
                    + "; rays are generated on device",
            "config": {"workload": f"{args.config.upper()}: {scene_name} {W}x{H}, {args.spp} spp/frame x {args.steps} frames, "
                                   f"{args.depth} bounces, wavefront pipeline; step = 1 frame",
                       "pixel_sharding": f"8x8 tiles interleaved over {world} rank(s)" + (", RCCL reduce to rank 0" if world > 1 else ""),
                       "frames_in_flight": shape.frames_in_flight, "sample_groups": shape.sample_groups},
            "rays": rays_total, "paths": paths_total, "rays_per_path": round(mean_len, 4),
            "rounds": st.rounds, "device_ms_rank0": round(st.ms_total, 3),
            "bvh": {"triangles": info.n_tris, "nodes": info.n_nodes, "height": info.bvh_height,
                    "build_ms": round(info.build_ms, 3), "bvh4_nodes": info.n_wide_nodes,
                    "bvh4_builder": ["collapsed LBVH", "surface-area sweep (<= 2048 triangles, ePreferFastTrace)"][info.bvh4_builder],
                    "extend_variant": pt.EXTEND_NAMES.get(st.extend_variant, str(st.extend_variant))},
        }
        if rays_minmax:
            out["rays_per_rank_min_max"] = rays_minmax
        if ingest:
            out["ingest"] = ingest
        if args.config == "c4":
            out["bvh"].update({"instances": info.n_instances, "tlas_nodes": info.n_tlas_nodes, "tlas_build_wall_ms": round(tlas_ms, 3)})
        base = None
        if not args.no_cpu_baseline and world == 1:
            base, _, _ = cpu_baseline(arrays, scene_name, W, H, args.spp, args.depth, budget_s=args.cpu_budget_s,
                                      instances=pt.cornell_grid_instances() if args.config == "c4" else None)
        # traversal work per ray, counted by an instrumented build of the same kernel on the same
        # BVH4 (untimed extra frame): feeds the scene-gather term of the algorithmic bytes
        nodes_per_ray = tris_per_ray = 0.0
        node_occ = tri_occ = None
        frame0_rays_gpu = frame0_film_gpu = None
        if st.extend_variant != pt.EXTEND_FLAT:
            ctx.reset_stats()
            scratch = pt.Film(ctx, W, H)
            pt.render(scene, scratch, pt.default_params(frame=0, frame_count=1, flags=pt.FLAG_COUNT_VISITS, **common))
            cst = ctx.stats()
            nodes_per_ray = cst.nodes_visited / max(cst.rays, 1)
            tris_per_ray = cst.tris_tested / max(cst.rays, 1)
            node_occ = cst.nodes_visited / (64.0 * cst.node_steps) if cst.node_steps else None
            tri_occ = cst.tris_tested / (64.0 * cst.tri_steps) if cst.tri_steps else None
            frame0_rays_gpu = cst.rays
            frame0_film_gpu = scratch.read_f32()
            scratch.close()
        if flags and st.launches_extend and st.ms_extend > 0:
            # dominant kernel = k_extend (closest-hit traversal).  Algorithmic bytes: 40 B/ray; the
            # 36-triangle scene + LBVH are LDS-resident so there is no scene-gather term.
            # scene gather: counted only when the scene exceeds the 32 MiB of L2
            scene_bytes = info.device_bytes
            # one BVH4 node visit of the HBM variant = 64 B (4 fp16 child boxes 48 B + 4 child words 16 B; the
            # two-level kernel reads fp32 nodes, 128 B); one triangle = 36 B of positions
            node_bytes = 128.0 if args.config == "c4" else 64.0
            gather = (nodes_per_ray * node_bytes + tris_per_ray * 36.0) if scene_bytes > (32 << 20) else 0.0
            bytes_extend = BYTES_EXTEND + gather
            gbs = bytes_extend * st.rays / (st.ms_extend * 1e-3) / 1e9
            pipeline_bytes = (bytes_extend + BYTES_SHADE + BYTES_PER_PATH / mean_len) * st.rays
            traffic = None
            prof = os.path.join(REPO, "profiles", {"c2": "r01_pmc_extend.json", "c4": "r01_pmc_extend_c4.json",
                                                   "c5": "r01_pmc_extend_c5.json"}[args.config])
            pmc = None
            if os.path.exists(prof):
                try:   # PMC HBM bytes per ray of the same kernel (committed rocprofv3 run) x this run's rays per launch
                    pmc = json.load(open(prof))
                    traffic = round(pmc["hbm_bytes_per_ray"] * st.rays / st.launches_extend, 1)
                except Exception:
                    traffic = pmc = None
            out["roofline"] = {
                "bound": "hbm", "kernel": {1: "k_extend_flat", 2: "k_extend<lds>", 3: "k_extend<hbm>"}.get(st.extend_variant, "?"),
                "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "launches": st.launches_extend,
                "avg_launch_us": round(st.ms_extend * 1e3 / st.launches_extend, 3),
                "algorithmic_bytes_per_launch": round(bytes_extend * st.rays / st.launches_extend, 1),
                "algorithmic_bytes_per_ray": round(bytes_extend, 1),
                "gather": {"bvh4_nodes_per_ray": round(nodes_per_ray, 2), "tris_per_ray": round(tris_per_ray, 2),
                           "lane_occupancy_node_steps": round(node_occ, 3) if node_occ else None,
                           "lane_occupancy_triangle_steps": round(tri_occ, 3) if tri_occ else None,
                           "bytes_per_ray": round(gather, 1), "scene_device_bytes": scene_bytes,
                           "source": "device counters of the instrumented extend kernel (PT_FLAG_COUNT_VISITS), 1 extra frame"},
                "extend_ms": round(st.ms_extend, 3), "shade_ms": round(st.ms_shade, 3),
                "pmc_profile": None if not pmc else {   # from the committed rocprofv3 PMC passes of this kernel
                    "valu_busy_fraction": round(pmc["valu_busy_fraction"], 3),
                    "valu_wave_instr_per_64_rays": round(pmc["valu_wave_instr_per_64_rays"], 1),
                    "wait_any_fraction_of_wave_cycles": round(pmc["wait_any_fraction_of_wave_cycles"], 3),
                    "l2_hit_rate": round(pmc["l2_hit_rate"], 3), "hbm_bytes_per_ray": round(pmc["hbm_bytes_per_ray"], 1),
                    "source": os.path.relpath(prof, REPO)},
                "pipeline_algorithmic_GBps": round(pipeline_bytes / (st.ms_total * 1e-3) / 1e9, 2),
                "note": ("Cornell (<8 KB scene+BVH) never leaves SGPRs/LDS: extend is VALU-issue bound, HBM sees only "
                         "queue I/O; the HBM fraction is physically meaningful on config C5 (1M triangles) only")
                        if args.config == "c2" else
                        ("TLAS (10k instances) + BLAS fit in L2: traversal is VALU/latency bound, HBM sees queue I/O only"
                         if args.config == "c4" else
                         "scene + BVH4 = 118 MB > L2: every node/triangle fetch is a 128/48-B gather through L2/MALL/HBM"),
            }
        if base and not args.no_cpu_baseline and world == 1:
            cpu_rays, cpu_spp, cpu_img = base.pop("_rays"), base.pop("_spp"), base.pop("_img")
            out["cpu_baseline"] = base
            if cpu_spp == args.spp and frame0_rays_gpu is not None:
                # the oracle rendered exactly frame 0 of this workload (the whole image): exact ray counts and every
                # float of the film must agree (checker only: none of this is in the timed region)
                out["frame0_ray_count"] = {"gpu": frame0_rays_gpu, "cpu_oracle": cpu_rays, "equal": frame0_rays_gpu == cpu_rays}
                out["frame0_film_bit_exact"] = bool(frame0_film_gpu.tobytes() == cpu_img.tobytes())
        print(json.dumps(out), flush=True)

    film.close()
    scene.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
