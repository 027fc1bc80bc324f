#!/usr/bin/env python3
"""bench.py -- the headline benchmark: Mrays/s (and ms/frame) of the radiance loop on the Cornell
box at 1920x1080, 8 bounces (BASELINE.json metric), on N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its own N ranks,
                                                            one process per GPU, rendezvous on 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 8 --config c3                   (BASELINE config C3: 1024 spp = 32 steps, 8 GPUs, RCCL gather)

A *step* is one frame = one reference launch (traceRaysKHR(W,H,1), main.cpp:659): 32 samples per
pixel, <= 8 rays each, blended into the film (raygen.rgen:41-91).  K steps = 32*K spp; K=2 is
exactly config C2 (64 spp); the default K=16 (512 spp) lets the device keep 16 frames in flight.  The scene + LBVH are resident in HBM before the timed
region; the timed region is the K frames (all kernels: generate, extend, shade/compact, resolve)
plus, for N>1, the one RCCL collective that assembles the presented image on rank 0.  Rank 0 prints ONE JSON line.

N>1 shards the 8x8 pixel tiles of the SAME image over the ranks (config C3's decomposition), so
the total work per step is fixed: "scaling": "strong".

The timed region -- EXACTLY K steps between barrier + synchronize -- is repeated (--reps, default 5, and until the
repetitions add up to >= 1 s of GPU time): `value` / `ms_per_step` are the MEDIAN repetition, `value_min` / `value_max`
/ `values` the spread (a single 0.15 s sample moved by +-3 % from run to run on one box).

Beside the headline the default single-GPU run appends, OUTSIDE the timed region, the legs VERDICT r01 asked for:
  * `c2_exact`            K = 2 (= 64 spp, exactly BASELINE config C2) in one call;
  * `latency_ms_1frame`   the reference's own dispatch shape: one blocking pt_render per frame (main.cpp:647-685);
  * `roofline.valu_*`     the VALU-issue side of the Cornell traversal kernel from LIVE block counters;
  * `roofline.traffic`    HBM-side bytes per launch of that kernel, counted: the same frames twice more in a child process under
                          `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (`traffic_live`; --no-live-pmc: the committed record);
  * `roofline_c5`         the traversal kernel on the 1M-triangle soup (BASELINE config C5), the one config whose
                          scene does not fit LDS/L2: algorithmic bytes per ray x rays per launch / average launch
                          time / 8 TB/s, every factor measured in this run (`--config c5` runs it as the headline,
                          `--config c5x` an 8M-triangle soup that does not fit the 256 MiB Infinity Cache either);
  * `roofline_c4`         the two-level kernel on the 10 000-instance grid (BASELINE config C4, 8 frames) and
    `roofline_c5x`        the 8M-triangle soup (2 frames): same shape as `roofline_c5`, incl. the VALU wave-instructions per
                          64 rays and the active lanes per instruction from the live block counters;
  * `cpu_baseline`        the oracle on all host cores and on one, with the CPU model.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
# VALU issue peak of the chip in wave64 instructions per second: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles (a SIMD
# issues a wave64 VALU op over two passes of its 32 lanes at best: v_add/v_mul; fma-class ops take 4)
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2
# algorithmic bytes per ray of the wavefront pipeline (SURVEY.md section 8d / DESIGN.md section 7)
BYTES_EXTEND = 40.0    # read ray 28 (index-free dense queue: 24 + 4 slot id passed along), write hit 12
BYTES_SHADE = 104.0
BYTES_PER_PATH = 96.0

# VALU instructions of the blocks of the traversal kernels, counted in the ISA of the shipped library by
# scripts/isa_blocks.py (re-run it after any change to the kernel headers; the table names the revision it was read at).
# A launch's VALU wave-instructions = sum over blocks of (wave executions counted live by the instrumented instantiation
# of the same template, PT_FLAG_COUNT_VISITS) x (instructions of the block); the lanes inside those executions, also
# counted live, weighted the same way give the active lanes per VALU instruction.
_MODEL = os.path.join(REPO, "profiles", "isa_valu_model.json")
ISA_VALU_MODEL = json.load(open(_MODEL)) if os.path.exists(_MODEL) else None
# which PMC record (scripts/make_pmc_json.py) carries the HBM-side traffic of a config's traversal kernel
PMC_RECORD = "r06_pmc_extend_{config}.json"
PMC_RECORD_SHADE = "r06_pmc_shade_{config}.json"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def effective_cores():
    """-> (threads this process can really run at once, how that was found).  A container may see every hardware thread
    of the host (os.cpu_count() = 256 on the GPU boxes) and still be capped by a cgroup CPU quota (cpu.max = 16 CPUs there):
    beyond the quota more threads only get throttled (scripts/probe_cpu_scaling.py: linear to 16 threads, erratic above)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = f"{n} schedulable hardware threads"
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:                                                                  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        how = f"cgroup CPU quota of {quota:g} CPUs on a host with {n} hardware threads"
        n = max(1, int(quota + 0.999))
    return n, how


def cpu_baseline(arrays, name, width, height, spp_max, depth, budget_s=10.0, instances=None):
    """The oracle (a CPU port of the reference shaders + software LBVH) timed on this host, on a
    bounded sample of the same workload: the same image, all cores, as many samples per pixel as fit
    into ~budget_s seconds (calibrated with a 1-spp pass, at most spp_max); then ONE thread on every 16th tile of it."""
    from oracle import pt_oracle as orc
    osc = orc.Scene(*arrays)
    if instances is not None:
        osc.set_instances(instances)      # two-level scene (config C4): the oracle walks its own TLAS
    cores, cores_how = effective_cores()
    t0 = time.perf_counter()
    osc.render_frame(orc.default_params(width=width, height=height, spp_per_frame=1, max_depth=depth), mode=1, nthreads=cores)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(spp_max, budget_s / max(t1, 1e-3))))
    p = orc.default_params(width=width, height=height, spp_per_frame=spp, max_depth=depth)
    t0 = time.perf_counter()
    img, rays, cnt, _ = osc.render_frame(p, mode=1, nthreads=cores)
    dt = time.perf_counter() - t0
    # single thread: every 16th 16x16 tile of the WHOLE image (row-major tile index % 16 == 0 -- 510 of the 8 160 tiles of a
    # 1080p frame: 120 tiles per row, so the subset walks down the columns 0, 16, 32 ... 112 of tiles and takes border and
    # centre in the image's own proportion): the same ray population as the all-core run, so `scaling_efficiency` compares
    # like with like (round 3 timed the single thread on the central crop, which has none of the cheap border pixels, and read 1.18)
    stride = max(16, cores)   # (more cores than 16: a thinner subset, so this leg takes about as long as the all-core one)
    p1 = orc.default_params(width=width, height=height, spp_per_frame=1, max_depth=depth)
    t0 = time.perf_counter()
    _, r1 = orc.render_tile_subset(osc, p1, stride, 0, mode=1, nthreads=1)
    d1 = time.perf_counter() - t0
    # the SAME samples per pixel as the all-core leg (round 4 gave the single thread a quarter of the budget: 15 spp against 32, and read
    # an "efficiency" of 1.02): on `cores` = 16 threads this leg then takes about as long as the all-core one
    spp1 = spp
    if spp1 > 1:
        p1 = orc.default_params(width=width, height=height, spp_per_frame=spp1, max_depth=depth)
        t0 = time.perf_counter()
        _, r1 = orc.render_tile_subset(osc, p1, stride, 0, mode=1, nthreads=1)
        d1 = time.perf_counter() - t0
    all_mrays, one_mrays = rays / dt / 1e6, r1 / d1 / 1e6
    base = {"value": round(all_mrays, 3), "unit": "Mrays/s", "cores": cores, "kind": "port", "_rays": rays, "_spp": spp, "_img": img,
            "cpu_model": cpu_model(), "cores_available": cores_how, "single_thread_mrays": round(one_mrays, 4),
            # all threads against `cores` x the single thread on the same ray population (every 16th tile of the same image, the same
            # samples); a core that runs alone clocks higher than sixteen together, so this is a lower bound of the software's scaling
            "scaling_efficiency": round(all_mrays / (cores * one_mrays), 4),
            # SURVEY 8d's gather term as written there: the instrumented ORACLE's walk of its own binary LBVH (32-B nodes, 36-B triangles)
            "oracle_walk": {"nodes_visited_per_ray": round(cnt.nodes_visited / max(rays, 1), 3), "tris_tested_per_ray": round(cnt.tris_tested / max(rays, 1), 3),
                            "gather_bytes_per_ray_8d": round((cnt.nodes_visited * 32.0 + cnt.tris_tested * 36.0) / max(rays, 1), 1),
                            "source": "oracle/pt_oracle.c counters of the all-core render above (binary LBVH, one primitive per leaf)"},
            "sample": f"{name} {width}x{height}, {spp} spp (frame 0), depth {depth}: {rays} rays in "
                      f"{dt:.2f} s; oracle/pt_oracle.c, software LBVH, gcc -O3 -march=native -ffp-contract=off, {cores} threads "
                      f"pulling 16x16 tiles from one counter; single thread: every {stride}th 16x16 tile of the same image, {spp1} spp, "
                      f"{r1} rays in {d1:.2f} s"}
    return base, cnt.nodes_visited / max(rays, 1), cnt.tris_tested / max(rays, 1)


def extend_kernel_name(pt, st, info, config):
    if info.n_instances:
        return "k_extend_inst16"
    lds = "k_extend<lds>"
    if info.n_wide_nodes <= 8191 and info.n_tris <= 2047:     # the compact no-spill instantiations (plan_extend)
        lds = "k_extend_lds7p"
    return {2: lds, 3: "k_extend<hbm>", 4: "k_extend8"}.get(st.extend_variant, "?")


def count_visits(pt, ctx, scene, W, H, common, frames=1, frame0=False):
    """The same `frames` frames once more, untimed, through the instrumented instantiation of the same traversal kernel
    (the wave-level block counts depend on how full the queues are, so the shape has to be the timed one: counted on a
    single frame the Cornell kernel shows 1075 VALU instructions per 64 rays, on the 16 of the timed run 930, which is what
    SQ_INSTS_VALU measures there); then, with `frame0` (the run that also times the CPU oracle), frame 0 alone through the
    shipped kernels for the film / ray-count comparison."""
    scratch = pt.Film(ctx, W, H)
    ctx.reset_stats()
    # (an instrumented render walks every ray unless pt_tuning.cull = 1 says otherwise: here the counts have to be those of the rays the TIMED
    # kernels walk -- cst.rays still holds every ray, walked_rays(cst) the walked ones)
    old = ctx.set_tuning(cull=1)
    try:
        pt.render(scene, scratch, pt.default_params(frame=0, frame_count=frames, flags=pt.FLAG_COUNT_VISITS, **common))
    finally:
        ctx.set_tuning(**old)
    cst = ctx.stats()
    film, rays0 = None, None
    if frame0:
        scratch.clear()
        ctx.reset_stats()
        pt.render(scene, scratch, pt.default_params(frame=0, frame_count=1, **common))
        rays0 = ctx.stats().rays
        film = scratch.read_f32()
    scratch.close()
    return cst, film, rays0


def valu_model(cst, kernel):
    """VALU wave-instructions per 64 rays of a traversal kernel, and the active lanes per VALU instruction, from live
    wave-level block counts x the blocks' instruction counts in the shipped ISA."""
    m = (ISA_VALU_MODEL or {}).get(kernel)
    if not m or not cst.wave_iterations:
        return None
    total = lanes = 0.0
    counts = {}
    for name, b in m["blocks"].items():
        waves = getattr(cst, b["waves"])
        ln = 64.0 * waves if b["lanes"] is None else float(getattr(cst, b["lanes"]))
        total += waves * b["valu"]
        lanes += ln * b["valu"]
        counts[name] = {"waves": waves, "lanes_per_wave": round(ln / waves, 1) if waves else None, "valu": b["valu"]}
    return {"wave_instr": total, "per_64_rays": total / max(walked_rays(cst), 1) * 64.0, "lanes_per_instr": lanes / max(total, 1.0),
            "revision": m.get("revision"), "blocks": counts}


def walked_rays(st):
    """Rays of a render that a kernel walked: pt_stats.rays less the camera rays of pixels that cannot see the scene, which every pipeline finishes
    without a walk (pt_stats.rays_culled; counted in `value` because the reference traces them, raygen.rgen:62).  The roofline blocks price only
    these, with per-ray averages of an instrumented run that walks the same rays (count_visits)."""
    return st.rays - int(getattr(st, "rays_culled", 0) or 0)


def roofline_block(pt, st, cst, info, config, note):
    """`roofline` for the dominant kernel (the closest-hit traversal) of a leg: every number from this run."""
    nodes_per_ray = cst.nodes_visited / max(walked_rays(cst), 1)   # (per WALKED ray: count_visits instruments the walked rays only)
    tris_per_ray = cst.tris_tested / max(walked_rays(cst), 1)
    node_occ = cst.nodes_visited / (64.0 * cst.node_steps) if cst.node_steps else None
    leaf_occ = cst.leaf_lanes / (64.0 * cst.tri_steps) if cst.tri_steps else None
    scene_bytes = info.device_bytes
    # one BVH4 node visit of the HBM variant = 64 B (4 fp16 child boxes 48 B + 4 child words 16 B); one triangle = 36 B of
    # positions.  Counted only when the scene exceeds the 32 MiB of L2; smaller scenes are LDS / L2 resident
    # and HBM sees the queue I/O only.
    node_bytes = 64.0
    gather = (nodes_per_ray * node_bytes + tris_per_ray * 36.0) if scene_bytes > (32 << 20) else 0.0
    bytes_extend = BYTES_EXTEND + gather
    gbs = bytes_extend * walked_rays(st) / (st.ms_extend * 1e-3) / 1e9
    kernel = extend_kernel_name(pt, st, info, config)
    r = {
        "bound": "hbm", "kernel": kernel,
        "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
        "traffic": None,
        "launches": st.launches_extend, "rays_per_launch": round(walked_rays(st) / st.launches_extend, 1),
        "rays_culled_per_launch": round((st.rays - walked_rays(st)) / st.launches_extend, 1),   # (finished by k_generate, never queued: not priced here)
        "avg_launch_us": round(st.ms_extend * 1e3 / st.launches_extend, 3),
        "algorithmic_bytes_per_launch": round(bytes_extend * walked_rays(st) / st.launches_extend, 1),
        "algorithmic_bytes_per_ray": round(bytes_extend, 1),
        "gather": {"bvh_nodes_per_ray": round(nodes_per_ray, 2), "node_bytes": node_bytes, "tris_per_ray": round(tris_per_ray, 2),
                   "bytes_per_ray": round(gather, 1), "scene_device_bytes": scene_bytes,
                   "source": "device counters of the instrumented extend kernel (PT_FLAG_COUNT_VISITS), the timed frames once more, untimed"},
        # lanes of a wave64 inside one node step / one leaf step (a leaf = one triangle, or a fan pair tested together)
        "active_lanes": {"node_steps": round(64 * node_occ, 1) if node_occ else None,
                         "leaf_steps": round(64 * leaf_occ, 1) if leaf_occ else None},
        "extend_ms": round(st.ms_extend, 3), "shade_ms": round(st.ms_shade, 3),
        "note": note,
    }
    # HBM-side traffic of the same kernel: PMC counters cannot be read from inside this process, so the figure is the
    # bytes per ray of this round's committed rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, corrected as
    # calibrated on known byte counts: profiles/r03_fetch_size_calibration.json) x this run's rays per launch
    prof = os.path.join(REPO, "profiles", PMC_RECORD.format(config=config))
    if os.path.exists(prof):
        try:
            pmc = json.load(open(prof))
            r["traffic"] = round(pmc["hbm_bytes_per_ray"] * walked_rays(st) / st.launches_extend, 1)
            # the counter side of `frac`: counted HBM-side bytes (2 x FETCH_SIZE + WRITE_SIZE: MALL hits included) instead of
            # algorithmic ones over the same launch time -- what "rocprof HBM GB/s against the chip's peak" reads
            r["frac_counted"] = round(pmc["hbm_bytes_per_ray"] * walked_rays(st) / (st.ms_extend * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            r["pmc_profile"] = {k: (round(pmc[k], 3) if isinstance(pmc[k], float) else pmc[k]) for k in
                                ("hbm_bytes_per_ray", "hbm_read_requests_per_ray", "valu_issue_frac", "valu_wave_instr_per_64_rays",
                                 "valu_active_lanes_per_instr", "wait_any_fraction_of_wave_cycles", "l2_hit_rate", "rocprof_avg_launch_us") if k in pmc}
            r["pmc_profile"]["source"] = os.path.relpath(prof, REPO)
        except Exception:
            pass
    vm = valu_model(cst, kernel)
    if vm:
        rays_per_s = walked_rays(st) / (st.ms_extend * 1e-3)   # the kernel's own rate (its launches overlap the other pipeline's shade)
        r["valu_wave_instr_per_64_rays"] = round(vm["per_64_rays"], 1)
        r["valu_active_lanes_per_instr"] = round(vm["lanes_per_instr"], 1)
        r["valu_frac"] = round(vm["per_64_rays"] / 64.0 * rays_per_s / VALU_PEAK_WAVE_INSTR, 4)
        r["valu_model"] = {"peak_wave_instr_per_s": VALU_PEAK_WAVE_INSTR, "isa_revision": vm["revision"], "blocks": vm["blocks"],
                           "source": "live wave-level block counts (PT_FLAG_COUNT_VISITS) x VALU instructions per block of the shipped ISA "
                                     "(profiles/isa_valu_model.json, scripts/isa_blocks.py)"}
    return r


def sum_counter_csvs(directory, ctr, prefixes):
    """rocprofv3 `*counter_collection.csv` files under `directory` -> ({prefix: sum of `ctr` over the dispatches of kernels whose short name starts with
    it}, {prefix: number of those dispatches}); the instrumented instantiations (PT_FLAG_COUNT_VISITS frames of the same run) do not count."""
    import csv, glob, re
    counting = re.compile(r"k_extend<\w+, true|k_extend_inst(16)?<true|k_extend8<true")
    tot = {p: 0.0 for p in prefixes}
    ids = {p: set() for p in prefixes}
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != ctr:
                continue
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            for p in prefixes:
                if k.startswith(p) and not counting.match(k):
                    tot[p] += float(row["Counter_Value"])
                    ids[p].add(row["Dispatch_Id"])
    return tot, {p: len(v) for p, v in ids.items()}


_LIVE_PMC_FAILED = [False]   # a pass that failed or timed out once is not tried again in this run (the legs would each wait for it)


def live_traffic(argv_child, prefixes=("k_extend", "k_shade"), timeout_s=150):
    """HBM-side bytes per ray of the frame's kernels MEASURED for this build on this box: the same command, short, twice under
    `rocprofv3 --kernel-trace --pmc` -- FETCH_SIZE and WRITE_SIZE in separate passes, nothing but the kernel trace beside them,
    as /opt/skills/guides/MI355X_MICROARCH.md prescribes -- corrected as calibrated (2 x FETCH_SIZE KiB + WRITE_SIZE KiB:
    profiles/r03_fetch_size_calibration.json).  -> {kernel prefix: record} or None (no rocprofv3, a pass failed or timed out: the
    caller keeps the committed record and says so)."""
    import shutil, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe or _LIVE_PMC_FAILED[0] or os.environ.get("ROCPROFILER_LIBRARY_CTOR") or "rocprofiler-sdk" in os.environ.get("LD_PRELOAD", ""):
        return None   # (no profiler, or this process is itself being profiled: no profiler inside a profiler)
    _LIVE_PMC_FAILED[0] = True   # until both passes have come back
    kib = {p: {} for p in prefixes}
    disp = {p: {} for p in prefixes}
    child = None
    t0 = time.perf_counter()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pt_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "-f", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__)] + argv_child,
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=timeout_s, env=dict(os.environ, TMPDIR="/tmp"))
            lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
            if r.returncode != 0 or not lines:
                return None
            child = json.loads(lines[-1])
            tot, n = sum_counter_csvs(d, ctr, prefixes)
            for p in prefixes:
                kib[p][ctr], disp[p][ctr] = tot[p], n[p]
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # both passes ran the same launches; the child's own line says how many rays they carried (a shade launch handles the rays of
    # the extend launch before it)
    launches = child["roofline"]["launches"] if "roofline" in child and child["roofline"].get("launches") else None
    if not launches:
        return None
    rays_per_launch = child["rays"] / launches
    out = {}
    for p in prefixes:
        if not disp[p].get("FETCH_SIZE") or not disp[p].get("WRITE_SIZE"):
            continue
        rd = 2.0 * 1024.0 * kib[p]["FETCH_SIZE"] / disp[p]["FETCH_SIZE"]
        wr = 1024.0 * kib[p]["WRITE_SIZE"] / disp[p]["WRITE_SIZE"]
        out[p] = {"hbm_bytes_per_ray": (rd + wr) / rays_per_launch, "hbm_read_bytes_per_ray": rd / rays_per_launch, "hbm_write_bytes_per_ray": wr / rays_per_launch,
                  "launches_profiled": disp[p]["FETCH_SIZE"], "rays_per_launch_profiled": round(rays_per_launch, 1), "seconds": round(time.perf_counter() - t0, 1),
                  "source": "live: this command under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) on this box"}
    _LIVE_PMC_FAILED[0] = not out
    return out or None


def apply_live_traffic(r_, lt, st, ms_kernel, launches):
    """`traffic` / `frac_counted` of a roofline block from a live_traffic() record (or the note that there is none)."""
    if lt:
        r_["traffic"] = round(lt["hbm_bytes_per_ray"] * st.rays / max(launches, 1), 1)
        if ms_kernel:
            r_["frac_counted"] = round(lt["hbm_bytes_per_ray"] * st.rays / (ms_kernel * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        r_["traffic_live"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in lt.items()}
    else:
        r_["traffic_live"] = {"source": "not measured in this run (no rocprofv3, or a pass failed): `traffic` is the committed record's bytes per ray x this run's rays"}


def roofline_shade_block(st, config):
    """`roofline_shade`: the other kernel of a wavefront round -- closest-hit shading, bounce, regeneration, compaction
    (closesthit.rchit:50-65, raygen.rgen:76-83) -- priced like the traversal kernel: SURVEY 8d's 104 algorithmic bytes per ray
    (read queue index 4 + hit 12 + path state 32, write next ray 24 + state 28 + index 4) x the rays of a launch / its average
    duration / 8 TB/s, every factor from this run's per-launch events; `traffic` = the counted HBM bytes per ray of the
    committed PMC pass x this run's rays per launch, like the traversal kernel's."""
    if not st.launches_shade or not st.ms_shade:
        return None
    rays_per_launch = walked_rays(st) / st.launches_shade
    avg_us = st.ms_shade * 1e3 / st.launches_shade
    gbs = BYTES_SHADE * rays_per_launch / (avg_us * 1e-6) / 1e9
    r = {"bound": "hbm", "kernel": "k_shade", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
         "traffic": None, "launches": st.launches_shade, "rays_per_launch": round(rays_per_launch, 1), "avg_launch_us": round(avg_us, 3),
         "algorithmic_bytes_per_ray": BYTES_SHADE, "algorithmic_bytes_per_launch": round(BYTES_SHADE * rays_per_launch, 1),
         # the same bytes over the device time of the whole timed region (the launches of the pipelines overlap)
         "frac_all_launches_over_device_time": round(BYTES_SHADE * walked_rays(st) / (st.ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
         "pipelines": st.pipelines, "shade_ms": round(st.ms_shade, 3),
         "note": "memory-latency bound: dense coalesced queue streams in and out + the scattered radiance term log; see pmc_profile"}
    prof = os.path.join(REPO, "profiles", PMC_RECORD_SHADE.format(config=config))
    if os.path.exists(prof):
        try:
            pmc = json.load(open(prof))
            r["traffic"] = round(pmc["hbm_bytes_per_ray"] * rays_per_launch, 1)
            r["frac_counted"] = round(pmc["hbm_bytes_per_ray"] * rays_per_launch / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
            r["pmc_profile"] = {k: (round(pmc[k], 4) if isinstance(pmc[k], float) else pmc[k]) for k in
                                ("hbm_bytes_per_ray", "hbm_write_bytes_per_ray", "valu_issue_frac", "valu_wave_instr_per_64_rays",
                                 "valu_active_lanes_per_instr", "wait_any_fraction_of_wave_cycles", "l2_hit_rate", "utcl1_miss_rate",
                                 "l1_to_l2_read_latency_cycles", "l1_to_l2_write_latency_cycles", "rocprof_avg_launch_us") if k in pmc}
            r["pmc_profile"]["source"] = os.path.relpath(prof, REPO)
        except Exception:
            pass
    return r


def wavefront_roofline_blocks(pt, st, cst, info, config, note, mean_len):
    """(`roofline` of the traversal kernel, `roofline_shade`) of a timed wavefront render: every factor from that render's per-launch events."""
    r = roofline_block(pt, st, cst, info, config, note)
    bytes_extend = r["algorithmic_bytes_per_ray"]
    pipeline_bytes = (bytes_extend + BYTES_SHADE + BYTES_PER_PATH / mean_len) * walked_rays(st)
    r["pipeline_algorithmic_GBps"] = round(pipeline_bytes / (st.ms_total * 1e-3) / 1e9, 2)
    # SURVEY 8d's canonical whole-pipeline figure: (extend + 104 shade + 96 per path / mean length) B per ray over the
    # device time of the timed region, of the HBM peak
    r["pipeline_frac"] = round(pipeline_bytes / (st.ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    # `frac` prices ONE launch against its own duration, and the launches of the concurrent pipelines share the chip:
    # with three pipelines a launch carries a third of the rays and lasts about as long as one of two did.  The same
    # algorithmic bytes over the device time of the timed region do not depend on how the work is cut into launches
    r["pipelines"] = st.pipelines
    r["frac_all_launches_over_device_time"] = round(bytes_extend * walked_rays(st) / (st.ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    return r, roofline_shade_block(st, config)


PMC_RECORD_FUSED = {"k_fused": "r06_pmc_fused_c2.json", "k_fused_inst": "r06_pmc_fused_c4.json"}


def fused_block_model(pt, ctx, scene, W, H, frames, spp, depth, rank=0, world=1):
    """Where k_fused's VALU instructions go, from THIS run: the same `frames` frames once more, untimed, through the kernel's instrumented twin
    (PT_FLAG_COUNT_VISITS on PT_PIPELINE_FUSED: wave executions and lanes of every block of its loop, include/pt_api.h pt_fused_block) x the blocks'
    VALU instruction counts in the shipped ISA (profiles/isa_valu_model.json "k_fused": scripts/isa_regions.py attributes every instruction of the
    product kernel's listing to a block by its .loc).  -> None when the scene / build has no instrumented form."""
    model = (ISA_VALU_MODEL or {}).get("k_fused")
    if not model or not hasattr(ctx, "block_counts"):
        return None
    film = pt.Film(ctx, W, H)
    old = ctx.set_tuning(cull=1)   # (an instrumented render walks every ray unless told otherwise: count what the timed kernel walks)
    try:
        ctx.reset_stats()
        pt.render(scene, film, pt.default_params(frame=0, frame_count=frames, width=W, height=H, spp_per_frame=spp, max_depth=depth, rank=rank, world=world,
                                                 pipeline=pt.PIPELINE_FUSED, flags=pt.FLAG_COUNT_VISITS))
        st, bc = ctx.stats(), ctx.block_counts()
    except Exception:
        return None
    finally:
        ctx.set_tuning(**old)
        film.close()
    walked = max(walked_rays(st), 1)
    rows, tot, lanes = {}, 0.0, 0.0
    for name, (waves, ln) in bc.items():
        v = model["blocks"].get(name, {}).get("valu", 0.0)
        tot += waves * v
        lanes += ln * v
        if waves:
            rows[name] = {"waves_per_64_rays": round(waves / walked * 64.0, 3), "lanes": round(ln / waves, 1), "valu": v, "valu_per_64_rays": round(waves * v / walked * 64.0, 1)}
    return {"per_64_rays": tot / walked * 64.0, "lanes_per_instr": lanes / max(tot, 1.0), "blocks": rows, "isa_revision": model.get("revision"),
            "tracing_lanes_per_pass": rows.get("TRACE", {}).get("lanes"),
            "source": "live: the timed frames once more through the instrumented twin (PT_FLAG_COUNT_VISITS, pt_get_block_counts) x VALU instructions per block of the "
                      "shipped ISA (profiles/isa_valu_model.json, scripts/isa_regions.py); fallback paths of the guarded three-FMA quotients priced at zero executions"}


def fused_roofline_block(st, mean_len, config, kernel, model=None):
    """`roofline` of the fused kernel (PT_PIPELINE_FUSED / what PT_PIPELINE_AUTO runs for scenes that live in LDS).  The kernel moves 0.2 B per ray
    through HBM; what binds it is VALU ISSUE, so that is the roofline the block states: `achieved` = wave64 VALU instructions per second
    -- the instructions per 64 walked rays from the live block model (fused_block_model; for the two-level kernel, which has no instrumented twin,
    the round's committed SQ_INSTS_VALU pass) x this run's walked rays per second of kernel time -- against `peak` = 256 CUs x 4 SIMDs x 2.4 GHz / 2
    cycles per wave64 op (an add / mul / mov / 2-source fma; min / max / compare / select / 3-source ops issue at ~4.3 cycles:
    profiles/r05q_valu_rate_ubench_sdwa.txt, so a real instruction mix saturates the SIMD well below 1.0).  `lanes` = active lanes per VALU
    instruction of 64.  The contract's form for a fused variant ("still divide by the ALGORITHMIC bytes of the wavefront design so
    designs are comparable") is kept as `frac_contract_8d` -- a comparison device, not traffic; `traffic` / `frac_counted` are what HBM really sees."""
    bytes_ray = BYTES_EXTEND + BYTES_SHADE + BYTES_PER_PATH / mean_len
    launches = max(st.launches_extend, 1)
    # rays the kernel WALKS: camera rays of pixels outside the scene box's projection are finished where their slot is handed out (pt_stats.rays_culled;
    # counted in `value` because the reference traces them, raygen.rgen:62) -- they gather no node and are priced at no byte and no instruction here
    culled = int(getattr(st, "rays_culled", 0))
    walked = st.rays - culled
    gbs = bytes_ray * walked / (st.ms_extend * 1e-3) / 1e9
    rays_per_s = walked / (st.ms_extend * 1e-3)
    r = {"bound": "valu_issue", "kernel": kernel,
         "variant": "fused: traversal and shading in one persistent kernel, path state in LDS / registers; HBM sees 16 B per slot (or per logged term)",
         "achieved": None, "peak": round(VALU_PEAK_WAVE_INSTR / 1e9, 1), "unit": "G wave64 VALU instr/s", "frac": None, "traffic": None,
         "lanes": None, "instr_per_64_rays": None,
         "launches": st.launches_extend, "rays_per_launch": round(st.rays / launches, 1), "rays_walked_per_launch": round(walked / launches, 1),
         "rays_culled_per_launch": round(culled / launches, 1), "avg_launch_us": round(st.ms_extend * 1e3 / launches, 3),
         "frac_contract_8d": round(gbs / HBM_PEAK_GBS, 5), "achieved_contract_8d_GBps": round(gbs, 2), "peak_contract_8d_GBps": HBM_PEAK_GBS,
         "algorithmic_bytes_per_ray": round(bytes_ray, 1), "algorithmic_bytes_per_launch": round(bytes_ray * walked / launches, 1),
         "note": "bound = VALU issue (the kernel walks LDS).  frac_contract_8d prices the WALKED rays by the wavefront design's algorithmic bytes (40 extend + 104 shade + "
                 "96 per path / mean path length) over the kernel's launch time against 8 TB/s, as SURVEY 8d prescribes for a fused variant: a comparison device -- the kernel "
                 "moves none of those bytes (traffic / frac_counted).  rays_culled_per_launch are camera rays of pixels that cannot see the scene, finished without a walk"}
    per64 = lanes = None
    src = None
    if model:
        per64, lanes, src = model["per_64_rays"], model["lanes_per_instr"], model["source"]
        r["blocks"] = model["blocks"]
        r["isa_revision"] = model["isa_revision"]
        r["tracing_lanes_per_pass"] = model["tracing_lanes_per_pass"]
    prof = os.path.join(REPO, "profiles", PMC_RECORD_FUSED.get(kernel, ""))
    pmc = None
    if os.path.isfile(prof):
        try:
            pmc = json.load(open(prof))
        except Exception:
            pmc = None
    if pmc:
        r["pmc_check"] = {"valu_wave_instr_per_64_rays": round(pmc["valu_wave_instr_per_64_rays"], 1), "valu_active_lanes_per_instr": round(pmc.get("valu_active_lanes_per_instr", 0.0), 1),
                          "source": os.path.relpath(prof, REPO) + " (SQ_INSTS_VALU / SQ_THREAD_CYCLES_VALU of the round's committed rocprofv3 pass: what the live model has to agree with)"}
        if pmc.get("salu_wave_instr_per_64_rays") is not None:  # the CU's one scalar unit beside the vector units
            r["pmc_check"]["salu_wave_instr_per_64_rays"] = round(pmc["salu_wave_instr_per_64_rays"], 1)
            r["pmc_check"]["salu_unit_frac"] = round(pmc.get("salu_unit_frac", 0.0), 3)
        if per64 is None:
            per64, lanes = pmc["valu_wave_instr_per_64_rays"], pmc.get("valu_active_lanes_per_instr")
            src = "SQ_INSTS_VALU per WALKED ray of " + os.path.relpath(prof, REPO) + " x this run's walked rays per second"
        r["traffic"] = round(pmc["hbm_bytes_per_ray"] * walked / launches, 1)
        r["frac_counted"] = round(pmc["hbm_bytes_per_ray"] * walked / (st.ms_extend * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
    if per64 is not None:
        r["instr_per_64_rays"] = round(per64, 1)
        r["lanes"] = round(lanes, 1) if lanes else None
        r["achieved"] = round(per64 / 64.0 * rays_per_s / 1e9, 2)
        r["frac"] = round(per64 / 64.0 * rays_per_s / VALU_PEAK_WAVE_INSTR, 4)
        r["lane_weighted_frac"] = round(r["frac"] * (lanes or 0.0) / 64.0, 4)
        r["instr_source"] = src
    return r


def wavefront_leg(pt, ctx, scene, info, W, H, args, config, note, child_argv):
    """The same K frames through PT_PIPELINE_WAVEFRONT (the north star's queue-per-bounce design) when the timed region ran the fused kernel:
    its rate and workspace, and the full roofline blocks of its two kernels from per-launch events of THIS run (+ live counter passes)."""
    import statistics
    kw = dict(width=W, height=H, spp_per_frame=args.spp, max_depth=args.depth, pipeline=pt.PIPELINE_WAVEFRONT)
    film = pt.Film(ctx, W, H)
    timed = pt.default_params(frame=0, frame_count=args.steps, flags=pt.FLAG_PROFILE, **kw)
    # what the queue design reaches when it may take the memory it wants (69 GB at 20 frames): the library's own 8 GB budget lifted for this leg,
    # the rate under that budget beside it
    at_default = None
    try:
        pt.render(scene, film, pt.default_params(frame=0, frame_count=args.steps, **kw))
        ctx.reset_stats()
        t0 = time.perf_counter()
        pt.render(scene, film, pt.default_params(frame=0, frame_count=args.steps, **kw))
        d0 = time.perf_counter() - t0
        s0 = ctx.stats()
        at_default = {"mem_budget_mb": "library default (8192)", "mrays_per_s": round(s0.rays / d0 / 1e6, 2), "workspace_bytes": s0.workspace_bytes,
                      "frames_in_flight": s0.frames_in_flight, "sample_groups": s0.sample_groups}
    except Exception as e:
        at_default = {"error": repr(e)}
    film.close()
    film = pt.Film(ctx, W, H)
    budget_before = ctx.set_tuning(mem_budget_mb=0)
    pt.render_prepare(scene, film, timed)
    shape = ctx.stats()
    kw.update(frames_in_flight=shape.frames_in_flight, sample_groups=shape.sample_groups)
    timed = pt.default_params(frame=0, frame_count=args.steps, flags=pt.FLAG_PROFILE, **kw)
    pt.render(scene, film, pt.default_params(frame=0, frame_count=max(args.warmup, 1), **kw))
    reps = []
    for _ in range(3):
        film.clear()
        ctx.reset_stats()
        t0 = time.perf_counter()
        pt.render(scene, film, timed)
        reps.append((time.perf_counter() - t0, ctx.stats()))
    reps.sort(key=lambda r: r[0])
    dt, st = reps[len(reps) // 2]
    film.close()
    cst, _, _ = count_visits(pt, ctx, scene, W, H, kw, args.steps)
    mean_len = st.rays / max(st.paths, 1)
    r, rs = wavefront_roofline_blocks(pt, st, cst, info, config, note, mean_len)
    if child_argv is not None:
        lt = None
        try:
            lt = live_traffic(child_argv + ["--pipeline", "wavefront"])
        except Exception:
            lt = None
        apply_live_traffic(r, (lt or {}).get("k_extend"), st, st.ms_extend, st.launches_extend)
        if rs:
            apply_live_traffic(rs, (lt or {}).get("k_shade"), st, st.ms_shade, st.launches_shade)
    ctx.set_tuning(**budget_before)
    return {"pipeline": "PT_PIPELINE_WAVEFRONT (generate / extend / shade queues, compacted per bounce)", "frames": args.steps,
            "mem_budget_mb": "none (lifted for this leg: pt_tuning.mem_budget_mb = 0)", "at_default_budget": at_default,
            "mrays_per_s": round(st.rays / dt / 1e6, 2), "values": [round(st.rays / x[0] / 1e6, 2) for x in reps], "ms_per_frame": round(dt * 1e3 / args.steps, 4),
            "rays": st.rays, "frames_in_flight": shape.frames_in_flight, "sample_groups": shape.sample_groups, "pipelines": st.pipelines,
            "workspace_bytes": st.workspace_bytes, "rounds": st.rounds, "roofline": r, "roofline_shade": rs}


def build_scene(pt, ctx, config, soup_tris, rank, bvh_quality):
    ingest = tlas_ms = None
    if config in ("c5", "c5x"):
        t0 = time.perf_counter()
        if config == "c5":
            obj = f"/tmp/pt_soup_{soup_tris}_rank{rank}_{os.getpid()}.obj"     # generated, not committed (139 MB of text)
            pt.write_soup_obj(obj, soup_tris, 1)
            t1 = time.perf_counter()
            arrays = pt.load_obj(obj)
            ingest = {"generate_s": round(t1 - t0, 3), "load_obj_s": round(time.perf_counter() - t1, 3), "obj_bytes": os.path.getsize(obj)}
            os.remove(obj)
            os.remove(obj[:-4] + ".mtl")
        else:
            arrays = pt.make_soup(soup_tris, 1)                  # the same soup without the OBJ text in between
            ingest = {"generate_s": round(time.perf_counter() - t0, 3), "source": "pth_make_soup (arrays, no OBJ text)"}
        name = f"soup {soup_tris} triangles (PCG seed 1)"
    else:
        arrays = pt.load_obj(pt.ASSET_CORNELL)
        name = "CornellBox-Original.obj"
    scene = pt.Scene(ctx, *arrays)          # upload + on-device LBVH build (untimed, reported apart)
    if bvh_quality == "fast_build":
        scene.set_bvh_quality(pt.BVH_PREFER_FAST_BUILD)
    if config == "c4":
        t0 = time.perf_counter()
        scene.set_instances(pt.cornell_grid_instances())      # TLAS build on device
        ctx.sync()
        tlas_ms = (time.perf_counter() - t0) * 1e3
        name = "CornellBox-Original.obj x 10 000 instances (100x100 grid, scale 0.009)"
    return scene, arrays, name, ingest, tlas_ms


NOTES = {
    "c2": "Cornell (<8 KB scene+BVH) never leaves LDS: extend is VALU-issue bound (valu_frac), HBM sees only queue I/O; "
          "the HBM fraction is physically meaningful on configs C5 / C5x only (roofline_c5, roofline_c5x)",
    "c4": "TLAS (10k instances) + BLAS fit in L2: traversal is VALU/latency bound, HBM sees queue I/O only",
    "c5": "scene + BVH4 = 160 MB > L2 (32 MiB) but < Infinity Cache (256 MiB): every node/triangle fetch is a 64-B gather "
          "through L1/L2/MALL; FETCH_SIZE counts MALL hits too",
    "c5x": "8M-triangle soup: the traversal working set (BVH4 64-B nodes + triangles) exceeds the 256 MiB Infinity Cache, "
           "gathers reach HBM",
}
NOTES["c3"] = NOTES["c2"]
LEG_SHAPE = {"c4": dict(spp=32, depth=8, tris=0), "c5": dict(spp=16, depth=16, tris=1000000), "c5x": dict(spp=16, depth=16, tris=8000000)}


def extra_leg(pt, ctx, W, H, config, frames, rank, live=False, oracle_walk=False):
    """The traversal kernel of another BASELINE config (C4 / C5 / C5x) in the default line: `frames` warm-up frames, then the
    same `frames` frames timed with per-launch events (identical launches, so rocprofv3's average over the whole process
    equals this leg's), then the same frames through the instrumented kernel for the visit and block counts."""
    sh = LEG_SHAPE[config]
    scene, arrays, name, ingest, tlas_ms = build_scene(pt, ctx, config, sh["tris"], rank, "fast_trace")
    info = scene.info()
    # what the workspace budget buys (the library plans within 8 GB unless the caller says otherwise): the leg's frames through the wavefront
    # pipeline's own shapes at 2 / 8 / 32 GB; the roofline block below is measured at 32 GB and says so
    by_budget = {}
    for mb in (2048, 8192, 32768):
        old_b = ctx.set_tuning(mem_budget_mb=mb)
        try:
            fb = pt.Film(ctx, W, H)
            pb = pt.default_params(frame=0, frame_count=frames, width=W, height=H, spp_per_frame=sh["spp"], max_depth=sh["depth"], pipeline=pt.PIPELINE_WAVEFRONT)
            pt.render(scene, fb, pb)
            ctx.reset_stats()
            t0 = time.perf_counter()
            pt.render(scene, fb, pb)
            db = time.perf_counter() - t0
            sb = ctx.stats()
            by_budget[f"{mb // 1024} GB" + (" (library default)" if mb == 8192 else "")] = {
                "mrays_per_s": round(sb.rays / db / 1e6, 1), "workspace_GB": round(sb.workspace_bytes / 2**30, 2), "frames_in_flight": sb.frames_in_flight, "sample_groups": sb.sample_groups}
            fb.close()
        except Exception as e:
            by_budget[f"{mb // 1024} GB"] = {"error": repr(e)}
        finally:
            ctx.set_tuning(**old_b)
    leg_budget = ctx.set_tuning(mem_budget_mb=32768)
    film = pt.Film(ctx, W, H)
    # (the wavefront pipeline explicitly: the leg is about its traversal kernel; what PT_PIPELINE_AUTO gives a caller of C4 is the `fused` block)
    common = dict(width=W, height=H, spp_per_frame=sh["spp"], max_depth=sh["depth"], pipeline=pt.PIPELINE_WAVEFRONT)
    timed = pt.default_params(frame=0, frame_count=frames, flags=pt.FLAG_PROFILE, **common)
    pt.render_prepare(scene, film, timed)
    shape = ctx.stats()
    common.update(frames_in_flight=shape.frames_in_flight, sample_groups=shape.sample_groups)
    timed = pt.default_params(frame=0, frame_count=frames, flags=pt.FLAG_PROFILE, **common)
    pt.render(scene, film, timed)              # warm-up: the same call
    film.clear()
    ctx.reset_stats()
    t0 = time.perf_counter()
    pt.render(scene, film, timed)
    dt = time.perf_counter() - t0
    st = ctx.stats()
    cst, _, _ = count_visits(pt, ctx, scene, W, H, common, frames)
    r = roofline_block(pt, st, cst, info, config, NOTES[config])
    out = {"workload": f"{config.upper()}: {name} {W}x{H}, {sh['spp']} spp/frame x {frames} frames, {sh['depth']} bounces",
           "mem_budget_mb": 32768, "by_budget": by_budget,
           "mrays_per_s": round(st.rays / dt / 1e6, 2), "ms_per_frame": round(dt * 1e3 / frames, 3), "rays": st.rays,
           "frames_in_flight": shape.frames_in_flight, "sample_groups": shape.sample_groups, "workspace_bytes": st.workspace_bytes,
           "bvh": {"triangles": info.n_tris, "bvh4_nodes": info.n_wide_nodes, "height": info.bvh_height, "build_ms": round(info.build_ms, 3),
                   "extend_variant": pt.EXTEND_NAMES.get(st.extend_variant)}}
    if ingest:
        out["ingest"] = ingest
    if config == "c4":
        out["bvh"].update({"instances": info.n_instances, "tlas_nodes": info.n_tlas_nodes, "tlas_build_wall_ms": round(tlas_ms, 3)})
        try:   # the same frames through PT_PIPELINE_FUSED's two-level kernel (csrc/fused_inst_kernel.h), opt-in like c2_fused
            fp = pt.default_params(frame=0, frame_count=frames, pipeline=pt.PIPELINE_FUSED, flags=pt.FLAG_PROFILE, width=W, height=H,
                                   spp_per_frame=sh["spp"], max_depth=sh["depth"])
            f2 = pt.Film(ctx, W, H)
            pt.render(scene, f2, fp)
            f2.clear()
            ctx.reset_stats()
            t1 = time.perf_counter()
            pt.render(scene, f2, fp)
            d1 = time.perf_counter() - t1
            s2 = ctx.stats()
            out["fused"] = {"pipeline": "PT_PIPELINE_FUSED (k_fused_inst): what PT_PIPELINE_AUTO runs for this scene", "mrays_per_s": round(s2.rays / d1 / 1e6, 2), "ms_per_frame": round(d1 * 1e3 / frames, 3),
                            "rays_equal_wavefront": s2.rays == st.rays, "film_equals_wavefront": bool(f2.read_f32().tobytes() == film.read_f32().tobytes()),
                            "frames_in_flight": s2.frames_in_flight, "sample_groups": s2.sample_groups, "workspace_bytes": s2.workspace_bytes,
                            "kernel_ms": round(s2.ms_extend, 3)}
            f2.close()
        except Exception as e:
            out["fused"] = {"error": repr(e)}
    if live:   # the leg's traffic counted like the headline's: the same frames in a child process under rocprofv3's counters
        lt = None
        try:
            lt = live_traffic(["--pmc-child", "--config", config, "--steps", str(frames), "--warmup", "0", "--reps", "1", "--no-cpu-baseline", "--no-extra-legs",
                               "--width", str(W), "--height", str(H), "--pipeline", "wavefront"], prefixes=("k_extend",))
        except Exception:
            lt = None
        apply_live_traffic(r["roofline"] if "roofline" in r else r, (lt or {}).get("k_extend"), st, st.ms_extend, st.launches_extend)
        # the same bytes over the DEVICE time of the leg (its pipelines' launches overlap, so the sum of launch durations exceeds it): what the
        # fabric carried for this kernel per second of the frame, of the 8 TB/s peak
        r["frac_all_launches_over_device_time"] = round(r["algorithmic_bytes_per_ray"] * walked_rays(st) / (st.ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        if lt and lt.get("k_extend"):
            r["frac_counted_over_device_time"] = round(lt["k_extend"]["hbm_bytes_per_ray"] * st.rays / (st.ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    if oracle_walk:
        # SURVEY 8d's gather term to the letter -- "sum over rays of nodesVisited x 32 B + trisTested x 36 B with counts taken from the instrumented
        # oracle traversing the same LBVH": the oracle's binary LBVH of the same triangles, walked by the oracle on every 64th 16x16 tile of the
        # same image at 1 spp (checker only, outside every timed region).  It prices a BINARY tree with one primitive per leaf, so it counts more
        # and smaller nodes than the 8-wide tree the kernel walks; both figures are given, the kernel's own tree stays the one in `frac`.
        try:
            from oracle import pt_oracle as orc
            t0 = time.perf_counter()
            osc = orc.Scene(*arrays)
            cores, _ = effective_cores()
            p1 = orc.default_params(width=W, height=H, spp_per_frame=1, max_depth=sh["depth"])
            orays, cnt = orc.render_tile_subset_counted(osc, p1, 64, 0, mode=1, nthreads=cores)
            g8d = (cnt.nodes_visited * 32.0 + cnt.tris_tested * 36.0) / max(orays, 1)
            gbs = (BYTES_EXTEND + g8d) * st.rays / (st.ms_extend * 1e-3) / 1e9
            r["gather_8d_literal"] = {"oracle_nodes_visited_per_ray": round(cnt.nodes_visited / max(orays, 1), 2), "oracle_tris_tested_per_ray": round(cnt.tris_tested / max(orays, 1), 2),
                                      "bytes_per_ray": round(g8d, 1), "algorithmic_bytes_per_ray": round(BYTES_EXTEND + g8d, 1), "achieved_GBps": round(gbs, 2),
                                      "would_be_frac": round(gbs / HBM_PEAK_GBS, 5), "oracle_rays_sampled": orays, "seconds": round(time.perf_counter() - t0, 2),
                                      "note": "not a bandwidth: the oracle walks a BINARY tree with one triangle per leaf (5x the node visits of the 8-wide tree), so pricing the "
                                              "kernel's time with those bytes can exceed the HBM peak -- a worse tree would score higher; the kernel's own tree stays in `frac`",
                                      "source": "oracle/pt_oracle.c counters, binary LBVH (32-B nodes, one triangle per leaf), every 64th 16x16 tile of the same image at 1 spp"}
            del osc
        except Exception as e:
            r["gather_8d_literal"] = {"error": repr(e)}
    out.update(r)
    ctx.set_tuning(**leg_budget)
    film.close()
    scene.close()
    return out


def fused_leg(pt, ctx, scene, film, W, H, spp, depth, steps, wavefront_mrays):
    """The same Cornell frames through PT_PIPELINE_FUSED (csrc/fused_kernel.h: the loop as ONE persistent kernel, the shape of the
    reference's own raygen shader), outside the timed region: K = steps, K = 2 (config C2 exactly) and one blocking call per frame,
    with the workspace each shape holds -- what the wavefront's 150 B per ray of queue traffic and tens of GB buy, and cost."""
    import statistics
    kw = dict(width=W, height=H, spp_per_frame=spp, max_depth=depth, pipeline=pt.PIPELINE_FUSED)
    out = {"pipeline": "PT_PIPELINE_FUSED", "wavefront_mrays_per_s_same_run": round(wavefront_mrays, 2)}
    for name, k in (("k_steps", steps), ("k2", 2)):
        p = pt.default_params(frame=0, frame_count=k, **kw)
        scratch = pt.Film(ctx, W, H)
        pt.render(scene, scratch, p)                  # allocates, warms up
        vals = []
        for _ in range(5):
            scratch.clear()
            ctx.reset_stats()
            t0 = time.perf_counter()
            pt.render(scene, scratch, p)
            d = time.perf_counter() - t0
            vals.append(ctx.stats().rays / d / 1e6)
        s_ = ctx.stats()
        out[name] = {"frames": k, "mrays_per_s": round(statistics.median(vals), 2), "min": round(min(vals), 2), "max": round(max(vals), 2),
                     "ms_per_frame": round(s_.rays / statistics.median(vals) / 1e3 / k, 3), "frames_in_flight": s_.frames_in_flight,
                     "sample_groups": s_.sample_groups, "workspace_bytes": s_.workspace_bytes, "rays": s_.rays}
        scratch.close()
    scratch = pt.Film(ctx, W, H)
    one = dict(frame_count=1, **kw)
    pt.render(scene, scratch, pt.default_params(frame=0, **one))
    lat = []
    ctx.reset_stats()
    for k in range(1, 9):
        t0 = time.perf_counter()
        pt.render(scene, scratch, pt.default_params(frame=k, **one))
        lat.append((time.perf_counter() - t0) * 1e3)
    s1 = ctx.stats()
    lat.sort()
    out["latency_1frame"] = {"median_ms": round(lat[len(lat) // 2], 3), "min_ms": round(lat[0], 3), "max_ms": round(lat[-1], 3),
                             "mrays_per_s": round(s1.rays / (sum(lat) * 1e-3) / 1e6, 2), "sample_groups": s1.sample_groups,
                             "workspace_bytes": s1.workspace_bytes}
    scratch.close()
    return out


def reference_dispatch_leg(pt, ctx, scene):
    """The reference's own dispatch, tested and timed: pt_params_default untouched -- 1024 x 1024 (main.cpp:16-17, 659), 32 spp (raygen.rgen:43),
    depth 8, PT_PIPELINE_AUTO -- (a) as the reference's frame loop issues it, one blocking pt_render per frame (pushConstants + traceRaysKHR + waitIdle,
    main.cpp:656-683), frames 1 .. 8 after a warm-up frame 0; (b) 20 frames in one call.  The full-size known answers of this launch (ray counts, film and
    bgra8 SHA-256 of frames 0 .. 3 from the oracle) are tests/golden/fullsize_hashes.json "ref1024" and the -m gpu test beside them."""
    import statistics
    p0 = pt.default_params()
    W, H = p0.width, p0.height
    film = pt.Film(ctx, W, H)
    pt.render(scene, film, pt.default_params(frame=0, frame_count=1))
    lat = []
    ctx.reset_stats()
    for k in range(1, 9):
        t0 = time.perf_counter()
        pt.render(scene, film, pt.default_params(frame=k, frame_count=1))
        lat.append((time.perf_counter() - t0) * 1e3)
    s1 = ctx.stats()
    out = {"workload": f"the reference's launch: {W}x{H} (main.cpp:16-17, 659), {p0.spp_per_frame} spp, {p0.max_depth} bounces, pt_params_default (PT_PIPELINE_AUTO -> "
                       f"{pt.PIPELINE_NAMES.get(s1.pipeline)})",
           "k1_blocking": {"median_ms": round(statistics.median(lat), 3), "min_ms": round(min(lat), 3), "max_ms": round(max(lat), 3), "frames": len(lat),
                           "mrays_per_s": round(s1.rays / (sum(lat) * 1e-3) / 1e6, 1), "mrays_per_s_walked": round(walked_rays(s1) / (sum(lat) * 1e-3) / 1e6, 1),
                           "rays_per_frame": s1.rays // len(lat), "sample_groups": s1.sample_groups, "tail_samples": s1.tail_samples, "workspace_bytes": s1.workspace_bytes}}
    film.clear()
    p20 = pt.default_params(frame=0, frame_count=20)
    pt.render(scene, film, p20)
    ts = []
    for _ in range(5):
        ctx.reset_stats()
        t0 = time.perf_counter()
        pt.render(scene, film, p20)
        ts.append(time.perf_counter() - t0)
    s20 = ctx.stats()
    d = statistics.median(ts)
    out["k20"] = {"ms_per_frame": round(d * 1e3 / 20, 3), "mrays_per_s": round(s20.rays / d / 1e6, 1), "mrays_per_s_walked": round(walked_rays(s20) / d / 1e6, 1),
                  "frames_in_flight": s20.frames_in_flight, "sample_groups": s20.sample_groups, "tail_samples": s20.tail_samples, "workspace_bytes": s20.workspace_bytes}
    film.close()
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, one process per GPU, exactly as
    torch.distributed.run would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment, rendezvous
    on 127.0.0.1), wait for them, and fail as a whole -- stopping the others by their PIDs -- if any rank fails (a rank that
    dies before a collective would otherwise leave its peers waiting in it).  Rank 0 prints the JSON line."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PT_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = set(range(n))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for o in live:
                    procs[o].terminate()
                deadline = time.time() + 10.0
                for o in list(live):
                    try:
                        procs[o].wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        procs[o].kill()
                        procs[o].wait()
                live.clear()
        time.sleep(0.05)
    return rc


def compact_line(out):
    """The ONE JSON line of the contract, small enough (<= 1.8 KB) that a log tail keeps it whole: the headline, the roofline of the dominant kernel, the CPU
    baseline, and the short form of every other leg.  The full blocks are printed before it as `# detail <name>: {json}` lines (not JSON lines themselves,
    so "the last JSON line" and "the only JSON line" are the same line) and written to gpurun_out/bench_last_full.json when that directory exists."""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None} if isinstance(d, dict) else None
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype") if k in out}
    c["data"] = "synthetic"
    w = (out.get("config") or {}).get("workload", "")
    c["config"] = {"workload": w.split("; step")[0].replace(" (the library default, PT_PIPELINE_AUTO)", " (library default)"), "pipeline": (out.get("config") or {}).get("pipeline")}
    for k in ("value_walked_only", "rays", "workspace_bytes", "frame0_film_bit_exact", "present_ms", "rccl_ranks"):
        if out.get(k) is not None:
            c[k] = out[k]
    if isinstance(out.get("selftest"), dict):
        c["selftest_ok"] = out["selftest"].get("ok")
    r = out.get("roofline")
    if isinstance(r, dict):
        c["roofline"] = pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "frac_contract_8d", "frac_counted", "lanes", "instr_per_64_rays",
                                 "launches", "avg_launch_us"))
        if c["roofline"].get("unit", "").startswith("G wave64"):
            c["roofline"]["unit"] = "Gwave-instr/s"
    b = out.get("cpu_baseline")
    if isinstance(b, dict):
        c["cpu_baseline"] = pick(b, ("value", "unit", "cores", "kind", "cpu_model", "single_thread_mrays"))
        if b.get("sample"):
            c["cpu_baseline"]["sample"] = str(b["sample"])[:70]
    if isinstance(out.get("c2_exact"), dict):
        c["c2_exact"] = pick(out["c2_exact"], ("mrays_per_s", "ms_total"))
    if isinstance(out.get("latency_1frame"), dict):
        c["latency_1frame"] = pick(out["latency_1frame"], ("median_ms",))
    rd = out.get("reference_dispatch")
    if isinstance(rd, dict) and "k1_blocking" in rd:
        c["reference_dispatch"] = {"size": "1024x1024x32spp", "k1_ms": rd["k1_blocking"].get("median_ms"), "k1_mrays": rd["k1_blocking"].get("mrays_per_s"),
                                   "k20_mrays": (rd.get("k20") or {}).get("mrays_per_s")}
    if isinstance(out.get("wavefront"), dict) and "mrays_per_s" in out["wavefront"]:
        c["wavefront"] = {"mrays": out["wavefront"]["mrays_per_s"], "extend_frac": (out.get("roofline_wavefront") or {}).get("frac"),
                          "shade_frac": (out.get("roofline_shade") or {}).get("frac")}
    for leg in ("c4", "c5", "c5x"):
        d = out.get("roofline_" + leg)
        if isinstance(d, dict) and "mrays_per_s" in d:
            e = {"mrays": d["mrays_per_s"], "kernel": d.get("kernel"), "frac": d.get("frac"), "frac_counted": d.get("frac_counted")}
            if isinstance(d.get("fused"), dict) and "mrays_per_s" in d["fused"]:
                e["fused_mrays"] = d["fused"]["mrays_per_s"]
                e["fused_GB"] = round(d["fused"].get("workspace_bytes", 0) / 2**30, 2)
            dflt = next((v for k, v in (d.get("by_budget") or {}).items() if "default" in k), None)
            if isinstance(dflt, dict) and "mrays_per_s" in dflt:
                e["at_8GB"] = dflt["mrays_per_s"]
            c[leg] = {k: v for k, v in e.items() if v is not None}
    return c


def emit(out, args):
    """Rank 0's output: the legs' full blocks first, one `# detail` line each, then the one JSON line of the contract (--full-line: the whole record as that line,
    for the dev scripts that read single keys of it)."""
    try:
        d = os.path.join(REPO, "gpurun_out")
        if os.path.isdir(d):
            json.dump(out, open(os.path.join(d, "bench_last_full.json"), "w"))
    except Exception:
        pass
    if args.full_line:
        print(json.dumps(out), flush=True)
        return
    big = ("roofline", "roofline_shade", "roofline_extend", "roofline_wavefront", "wavefront", "c2_exact", "latency_1frame", "reference_dispatch", "c2_fused",
           "roofline_c4", "roofline_c5", "roofline_c5x", "cpu_baseline", "config", "bvh", "selftest", "frame0_ray_count")
    for k in big:
        if out.get(k) is not None:
            print(f"# detail {k}: " + json.dumps(out[k]), flush=True)
    rest = {k: v for k, v in out.items() if k not in big}
    print("# detail headline: " + json.dumps(rest), flush=True)
    line = json.dumps(compact_line(out), separators=(",", ":"))
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="frames per timed region (default 16; --config c3: 32 = 1024 spp)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed region (median reported); more are run until they add up to >= 1 s")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5", "c5x"], default="c2",
                    help="c2 = Cornell box (the headline); c3 = the same, 1024 spp = 32 steps (BASELINE config C3: run it with --gpus 8); "
                         "c4 = Cornell x 10 000 instances (two-level BVH); c5 = 1M-triangle soup, 16 spp/frame, depth 16; "
                         "c5x = the same recipe with 8M triangles (> Infinity Cache)")
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--soup-tris", type=int, default=None)
    ap.add_argument("--frames-in-flight", type=int, default=0)
    ap.add_argument("--sample-groups", type=int, default=0)
    ap.add_argument("--extend", choices=["auto", "lds", "hbm", "hbm8"], default="auto", help="closest-hit kernel variant")
    ap.add_argument("--pipeline", choices=["auto", "wavefront", "fused"], default="auto",
                    help="auto = PT_PIPELINE_AUTO, what pt_params_default gives a caller: the fused kernel where the scene lives in LDS (c2 / c3 / c4), the "
                         "wavefront queues otherwise (c5 / c5x); wavefront = generate / extend / shade queues (the north star's design); fused = "
                         "PT_PIPELINE_FUSED, the whole loop as one persistent kernel")
    ap.add_argument("--bvh-quality", choices=["fast_trace", "fast_build"], default="fast_trace",
                    help="fast_trace = the reference's ePreferFastTrace (main.cpp:419, default); fast_build = collapsed LBVH only")
    ap.add_argument("--sort-rays", choices=["auto", "on", "off"], default="auto",
                    help="per-round device sort of the extend queue by (origin cell, octant); auto = scenes beyond the Infinity Cache")
    ap.add_argument("--selftest", action="store_true", default=True,
                    help="N > 1 (on by default): before any timing, present a rank-coloured film through the run's own collective and check on rank 0 that every "
                         "tile carries its owner's colour and that the communicator connected N ranks; the record goes into the line, a failure ends the run")
    ap.add_argument("--no-selftest", dest="selftest", action="store_false", help="N > 1: skip the presentation self-test")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not hipEvent-time each extend/shade launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed PMC record instead of two nested rocprofv3 passes of this command")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the nested run of live_traffic()
    ap.add_argument("--mem-budget-mb", type=int, default=None, help="pt_tuning.mem_budget_mb for the headline's context: the workspace budget the shapes are planned within "
                                                                    "(default: the library's own 8192; 0 = none).  The extra legs name their own budgets")
    ap.add_argument("--full-line", action="store_true", help="print the whole record as the one JSON line (the default prints the legs as `# detail` lines and a compact final line)")
    ap.add_argument("--no-extra-legs", action="store_true", help="headline only: no c2_exact / latency / roofline_c4 / _c5 / _c5x legs")
    ap.add_argument("--c5-frames", type=int, default=4, help="frames of the roofline_c5 leg")
    ap.add_argument("--c4-frames", type=int, default=8, help="frames of the roofline_c4 leg")
    ap.add_argument("--c5x-frames", type=int, default=2, help="frames of the roofline_c5x leg")
    ap.add_argument("--cpu-budget-s", type=float, default=10.0,
                    help="seconds of oracle time for cpu_baseline (it renders as many spp of frame 0 as fit; when that is the "
                         "whole frame, film and ray count are compared with the GPU's)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))    # no launcher: bench.py is its own

    # dmabuf IPC: what RCCL needs between processes on this driver (already exported on the pool's boxes; under a foreign launcher it may not be)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world                                      # the launcher's world size is authoritative
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU fallback)")
    # PT_BENCH_EMULATE=1 (dev check of the N > 1 code path on a 1-GPU box): every rank uses GPU 0 and the
    # collectives run over gloo on host copies.  Never used for reported numbers.
    emulate = os.environ.get("PT_BENCH_EMULATE") == "1" and world > 1
    if emulate:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank}, this node has {torch.cuda.device_count()} "
                 f"(--gpus {world}; PT_BENCH_EMULATE=1 runs all ranks on GPU 0 over gloo, for development only)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if emulate else dev     # where the collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emulate:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
    ptd = importlib.import_module("single-file-vulkan-pathtracing_amd.distributed")

    W, H = args.width, args.height
    scene_config = "c2" if args.config == "c3" else args.config
    if args.steps is None:
        args.steps = 32 if args.config == "c3" else 16
    if args.config in ("c5", "c5x"):
        args.spp = args.spp or 16
        args.depth = args.depth or 16
        args.soup_tris = args.soup_tris or (1000000 if args.config == "c5" else 8000000)
    else:
        args.spp = args.spp or 32
        args.depth = args.depth or 8
    stream = torch.cuda.current_stream(dev)
    ctx = pt.Context(local_rank, stream=stream.cuda_stream)
    if args.mem_budget_mb is not None:
        ctx.set_tuning(mem_budget_mb=args.mem_budget_mb)
    scene, arrays, scene_name, ingest, tlas_ms = build_scene(pt, ctx, scene_config, args.soup_tris, rank, args.bvh_quality)
    info = scene.info()
    film_t = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)   # this rank's accumulation film (torch owns the memory)
    film = pt.Film(ctx, W, H, device_ptr=film_t.data_ptr())
    presenter = ptd.Presenter(pt, ctx, film, film_t, rank, world, cdev, emulate) if world > 1 else None
    flags = 0 if args.no_kernel_events else pt.FLAG_PROFILE
    sort_flag = {"auto": 0, "on": pt.FLAG_SORT_RAYS, "off": pt.FLAG_NO_SORT_RAYS}[args.sort_rays]
    flags |= sort_flag
    common = dict(width=W, height=H, spp_per_frame=args.spp, max_depth=args.depth, rank=rank, world=world,
                  frames_in_flight=args.frames_in_flight, sample_groups=args.sample_groups,
                  pipeline={"auto": pt.PIPELINE_AUTO, "wavefront": pt.PIPELINE_WAVEFRONT, "fused": pt.PIPELINE_FUSED}[args.pipeline],
                  extend={"auto": pt.EXTEND_AUTO, "lds": pt.EXTEND_LDS, "hbm": pt.EXTEND_HBM, "hbm8": pt.EXTEND_HBM8}[args.extend])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # the workspace is sized once for the timed call's shape (frames in flight x sample groups) ...
    timed = pt.default_params(frame=0, frame_count=args.steps, flags=flags, **common)
    pt.render_prepare(scene, film, timed)
    shape = ctx.stats()
    ran = pt.PIPELINE_NAMES.get(shape.pipeline, str(shape.pipeline))     # what PT_PIPELINE_AUTO resolved to for this scene and call
    # (the shape the library chose is named in the timed call -- except a head + tail shape, which only the library's own rule gives: naming
    # sample_groups = 1 would turn it into the plain one-group shape)
    common.update(frames_in_flight=shape.frames_in_flight, sample_groups=0 if shape.tail_samples else shape.sample_groups, pipeline=shape.pipeline)
    timed = pt.default_params(frame=0, frame_count=args.steps, flags=flags, **common)
    # ... then W untimed warm-up frames run through the same kernels
    if args.warmup > 0:
        pt.render(scene, film, pt.default_params(frame=0, frame_count=args.warmup, flags=sort_flag, **common))
    selftest = None
    if presenter:      # communicator set-up (the first collective of a process) is not a step: always outside the timed region
        presenter.present()
        if args.selftest:
            selftest = presenter.selftest(film_t)
            verdict = torch.tensor([1 if (rank != 0 or selftest["ok"]) else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
            if not int(verdict.item()):
                if rank == 0:
                    print(json.dumps({"selftest": selftest, "error": "presentation self-test failed: tiles did not arrive from their owners"}), flush=True)
                sys.exit(2)

    # ---- the timed region: EXACTLY `steps` frames (+ the one collective that presents the image for N > 1) between
    # barrier + synchronize, repeated; the film is cleared and the counters reset between repetitions, outside of it
    reps = []       # (seconds, present seconds, stats) per repetition; seconds = MAX over ranks
    presented = film_t
    while True:
        film.clear()
        ctx.reset_stats()
        barrier()
        t0 = time.perf_counter()
        pt.render(scene, film, timed)          # blocking: returns when the device is done (main.cpp:683 waitIdle)
        t1 = time.perf_counter()
        if presenter:
            presented = presenter.present()    # the one collective per presented image (none for N = 1)
            torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        barrier()
        dt = time.perf_counter() - t0
        tp = t2 - t1
        if world > 1:
            tm = torch.tensor([dt, tp], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt, tp = float(tm[0].item()), float(tm[1].item())
        reps.append((dt, tp, ctx.stats()))
        # (every rank sees the same reduced times, so all ranks stop together)
        if len(reps) >= max(1, args.reps) and (sum(r[0] for r in reps) >= 1.0 or len(reps) >= 25):
            break
    order = sorted(range(len(reps)), key=lambda i: reps[i][0])
    med = order[len(order) // 2]            # the median repetition: its time, its counters, its per-kernel events
    dt, present_s, st = reps[med]
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
        values = [round(rays_total / r[0] / 1e6, 2) for r in reps]    # every repetition traces the same rays
        out = {
            "metric": {"c2": "Mrays/s, Cornell Box 1920x1080 @ 8 bounces (ms/frame in ms_per_step)",
                       "c3": "Mrays/s, Cornell Box 1920x1080 @ 8 bounces, 1024 spp progressive (ms/frame in ms_per_step)",
                       "c4": "Mrays/s, Cornell Box x 10k instances (two-level BVH) 1920x1080 @ 8 bounces (BASELINE config C4)",
                       "c5": "Mrays/s, 1M-triangle soup 1920x1080 @ 16 bounces (BASELINE config C5)",
                       "c5x": "Mrays/s, 8M-triangle soup (> Infinity Cache) 1920x1080 @ 16 bounces"}[args.config],
            "value": round(rays_total / dt / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 4),
            "ms_per_1spp_pass": round(dt * 1e3 / args.steps / args.spp, 5),   # also per sample-per-pixel pass
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("CornellBox-Original.obj (the reference's own scene, 36 triangles)" if scene_config in ("c2", "c4") else
                     "synthetic triangle soup (generator pth_write_soup_obj / pth_make_soup, seed 1)"
                     + (", written as OBJ+MTL and parsed by the host loader" if args.config == "c5" else ""))
                    + "; rays are generated on device",
            "config": {"workload": f"{args.config.upper()}: {scene_name} {W}x{H}, {args.spp} spp/frame x {args.steps} frames, "
                                   f"{args.depth} bounces, {ran} pipeline" + (" (the library default, PT_PIPELINE_AUTO)" if args.pipeline == "auto" else "") + "; step = 1 frame",
                       "pipeline": ran, "pipeline_requested": args.pipeline,
                       "pixel_sharding": f"8x8 tiles interleaved over {world} rank(s)" +
                                         (f", {presenter.describe()}" if presenter else ""),
                       "frames_in_flight": shape.frames_in_flight, "sample_groups": shape.sample_groups, "tail_samples": shape.tail_samples, "pipelines": st.pipelines,
                       "sort_rays": args.sort_rays,
                       "mem_budget_mb": args.mem_budget_mb if args.mem_budget_mb is not None else "library default (8192)"},
            # the timed region repeated: value / ms_per_step are the median repetition
            "reps": len(reps), "value_min": min(values), "value_max": max(values), "values": values,
            "timed_seconds_total": round(sum(r[0] for r in reps), 4),
            "rays": rays_total, "paths": paths_total, "rays_per_path": round(mean_len, 4),
            # (rank 0's share) camera rays of pixels outside the scene box's projection: counted in `value` -- the reference traces them, raygen.rgen:62 --
            # and finished by the fused kernel without a walk (pt_tuning.cull); value_walked_only prices the step by the walked rays alone
            "rays_culled_rank0": int(getattr(st, "rays_culled", 0)),
            "rays_note": "`rays` (and `value`) count every ray the reference dispatches (raygen.rgen:62), as the CPU oracle does; rays_culled_rank0 of them are camera rays of "
                         "pixels outside the projection of the scene's box, which every pipeline finishes as the misses they are without walking the tree "
                         "(bit-identical film; pt_tuning.cull = 0 walks them); value_walked_only = the same time priced by the walked rays alone; roofline blocks price walked rays only",
            "value_walked_only": round((st.rays - int(getattr(st, "rays_culled", 0))) / max(st.rays, 1) * rays_total / dt / 1e6, 2),
            "rounds": st.rounds, "device_ms_rank0": round(st.ms_total, 3),
            "workspace_bytes": st.workspace_bytes,
            "bvh": {"triangles": info.n_tris, "nodes": info.n_nodes, "height": info.bvh_height,
                    "build_ms": round(info.build_ms, 3), "bvh4_nodes": info.n_wide_nodes,
                    "bvh4_builder": ["collapsed LBVH", "surface-area sweep, one primitive per leaf (ePreferFastTrace)", "PLOC rebuild of the binary tree (ePreferFastTrace)"][min(info.bvh4_builder, 2)],
                    "extend_variant": pt.EXTEND_NAMES.get(st.extend_variant, str(st.extend_variant))},
        }
        if world > 1:
            out["rays_per_rank_min_max"] = rays_minmax
            out["rccl_ranks"] = presenter.ranks_seen
            out["present_ms"] = round(present_s * 1e3, 4)     # the collective + pack / unpack, max over ranks, median repetition
            out["launcher"] = "bench.py's own (one process per GPU)" if os.environ.get("PT_BENCH_SPAWNED") else "external (torch.distributed.run)"
            if selftest is not None:
                out["selftest"] = selftest
        # sum of the presented image (N = 1: the film; N > 1: what the gather assembled on rank 0), order-insensitive in float64
        out["presented_checksum"] = float(presented.double().sum().item())
        if ingest:
            out["ingest"] = ingest
        if args.config == "c4":
            out["bvh"].update({"instances": info.n_instances, "tlas_nodes": info.n_tlas_nodes, "tlas_build_wall_ms": round(tlas_ms, 3)})
        # traversal work per ray, counted by an instrumented build of the same kernel on the same
        # BVH4 (untimed extra frames): feeds the scene-gather term of the algorithmic bytes and the VALU model
        frame0_rays_gpu = frame0_film_gpu = None
        cst = None
        live = world == 1 and not (args.no_extra_legs or args.no_live_pmc or args.pmc_child)
        child_common = (["--mem-budget-mb", str(args.mem_budget_mb)] if args.mem_budget_mb is not None else []) + ["--pmc-child", "--config", args.config, "--steps", str(args.steps), "--warmup", "0", "--reps", "1", "--no-cpu-baseline", "--no-extra-legs",
                        "--width", str(W), "--height", str(H), "--spp", str(args.spp), "--depth", str(args.depth), "--extend", args.extend,
                        "--frames-in-flight", str(args.frames_in_flight), "--sample-groups", str(args.sample_groups), "--sort-rays", args.sort_rays,
                        "--bvh-quality", args.bvh_quality] + (["--soup-tris", str(args.soup_tris)] if args.soup_tris else [])
        if ran == "fused":   # (no instrumented form: frame 0 alone for the film / ray-count comparison)
            if not args.no_cpu_baseline and world == 1:
                scratch = pt.Film(ctx, W, H)
                ctx.reset_stats()
                pt.render(scene, scratch, pt.default_params(frame=0, frame_count=1, **common))
                frame0_rays_gpu, frame0_film_gpu = ctx.stats().rays, scratch.read_f32()
                scratch.close()
            if st.launches_extend and st.ms_extend > 0:
                kname = "k_fused_inst" if info.n_instances else "k_fused"
                model = fused_block_model(pt, ctx, scene, W, H, args.steps, args.spp, args.depth, rank, world) if kname == "k_fused" and not args.pmc_child else None
                out["roofline"] = fused_roofline_block(st, mean_len, scene_config, kname, model)
                if live:
                    lt = None
                    try:
                        lt = live_traffic(child_common + ["--pipeline", "fused"], prefixes=(kname,))
                    except Exception:
                        lt = None
                    apply_live_traffic(out["roofline"], (lt or {}).get(kname), st, st.ms_extend, st.launches_extend)
                    if out["roofline"].get("frac_counted") is not None:
                        out["roofline"]["frac_counted_note"] = "counted HBM-side bytes of the kernel over its launch time, of the 8 TB/s peak: what the fused kernel really moves"
            # the north star's own design beside it: the same frames through the wavefront queues, with the roofline blocks of ITS two kernels
            if world == 1 and not args.no_extra_legs and flags:
                try:
                    wl = wavefront_leg(pt, ctx, scene, info, W, H, args, scene_config, NOTES[args.config], child_common if live else None)
                    out["roofline_wavefront"] = wl.pop("roofline")
                    out["roofline_shade"] = wl.pop("roofline_shade")
                    out["wavefront"] = wl
                except Exception as e:
                    out["wavefront"] = {"error": repr(e)}
        else:
            cst, frame0_film_gpu, frame0_rays_gpu = count_visits(pt, ctx, scene, W, H, common, args.steps, frame0=not args.no_cpu_baseline and world == 1)   # (rank 0's shard when N > 1)
        if ran != "fused" and flags and st.launches_extend and st.ms_extend > 0 and cst is not None:
            out["roofline"], out["roofline_shade"] = wavefront_roofline_blocks(pt, st, cst, info, scene_config, NOTES[args.config], mean_len)
            # `roofline` is the kernel with the most time in the timed region; the other one keeps its block beside it
            if out["roofline_shade"] and st.ms_shade > st.ms_extend:
                out["roofline"], out["roofline_extend"] = dict(out["roofline_shade"], note_dominant="k_shade has the most kernel time of the timed region "
                                                               f"({st.ms_shade:.1f} ms of launches against the traversal kernel's {st.ms_extend:.1f}); the traversal kernel is in roofline_extend"), out["roofline"]
            if live:
                # roofline.traffic measured, not looked up: the same frames twice more in a child process under rocprofv3's counters
                lt = None
                try:
                    lt = live_traffic(child_common + ["--pipeline", "wavefront"])
                except Exception:
                    lt = None
                ext = out.get("roofline_extend") or out["roofline"]
                apply_live_traffic(ext, (lt or {}).get("k_extend"), st, st.ms_extend, st.launches_extend)
                shd = out["roofline"] if "roofline_extend" in out else out.get("roofline_shade")
                if shd:
                    apply_live_traffic(shd, (lt or {}).get("k_shade"), st, st.ms_shade, st.launches_shade)
        if world == 1 and not args.no_extra_legs and args.config in ("c2", "c3"):
            # ---- the reference's own dispatch shapes (outside the timed region) --------------------------------
            ctx.reset_stats()
            film.clear()
            exact = pt.default_params(frame=0, frame_count=2, width=W, height=H, spp_per_frame=args.spp, max_depth=args.depth)
            pt.render_prepare(scene, film, exact)
            pt.render(scene, film, exact)                          # warm-up of this shape
            film.clear()
            torch.cuda.synchronize(dev)
            d2s = []
            for _ in range(5):                                     # (frame 0 starts the film over: the same two frames five times, the median call)
                ctx.reset_stats()
                t0 = time.perf_counter()
                pt.render(scene, film, exact)
                d2s.append(time.perf_counter() - t0)
            s2 = ctx.stats()
            d2 = sorted(d2s)[len(d2s) // 2]
            out["c2_exact"] = {"workload": f"BASELINE config C2 exactly: {W}x{H}, 64 spp = 2 frames x 32, {args.depth} bounces, one pt_render",
                               "mrays_per_s": round(s2.rays / d2 / 1e6, 2), "ms_total": round(d2 * 1e3, 3), "ms_per_frame": round(d2 * 1e3 / 2, 3),
                               "ms_total_calls": [round(x * 1e3, 3) for x in d2s],
                               "rays": s2.rays, "frames_in_flight": s2.frames_in_flight, "sample_groups": s2.sample_groups, "tail_samples": s2.tail_samples,
                               "pipeline": pt.PIPELINE_NAMES.get(s2.pipeline) + " (PT_PIPELINE_AUTO, what pt_params_default gives a caller)"}
            one = dict(width=W, height=H, spp_per_frame=args.spp, max_depth=args.depth, frame_count=1)
            pt.render_prepare(scene, film, pt.default_params(frame=0, **one))
            film.clear()
            pt.render(scene, film, pt.default_params(frame=0, **one))
            lat = []
            ctx.reset_stats()
            for k in range(1, 9):                                  # one blocking pt_render per frame, as main.cpp:647-685
                t0 = time.perf_counter()
                pt.render(scene, film, pt.default_params(frame=k, **one))
                lat.append((time.perf_counter() - t0) * 1e3)
            s1 = ctx.stats()
            lat.sort()
            out["latency_ms_1frame"] = round(lat[len(lat) // 2], 3)
            out["latency_1frame"] = {"median_ms": round(lat[len(lat) // 2], 3), "min_ms": round(lat[0], 3), "max_ms": round(lat[-1], 3),
                                     "mrays_per_s": round(s1.rays / (sum(lat) * 1e-3) / 1e6, 2), "frames": len(lat),
                                     "sample_groups": s1.sample_groups, "tail_samples": s1.tail_samples, "workspace_bytes": s1.workspace_bytes,
                                     "pipeline": pt.PIPELINE_NAMES.get(s1.pipeline) + " (PT_PIPELINE_AUTO, what pt_params_default gives a caller)",
                                     "shape": "K = 1: one blocking pt_render per frame (pushConstants + traceRaysKHR + waitIdle, main.cpp:656-683)"}
            # ---- the reference's OWN launch: WIDTH = HEIGHT = 1024 (main.cpp:16-17), traceRaysKHR(1024, 1024, 1) (main.cpp:659), 32 spp, depth 8 -- exactly
            # what pt_params_default returns -- as the reference's host issues it (one blocking call per frame, main.cpp:647-685) and as one call of 20 frames
            try:
                out["reference_dispatch"] = reference_dispatch_leg(pt, ctx, scene)
            except Exception as e:
                out["reference_dispatch"] = {"error": repr(e)}
            if ran == "wavefront" and scene_config == "c2":
                try:
                    out["c2_fused"] = fused_leg(pt, ctx, scene, film, W, H, args.spp, args.depth, args.steps, st.rays / dt / 1e6)
                except Exception as e:
                    out["c2_fused"] = {"error": repr(e)}
            # ---- the other BASELINE configs' traversal kernels: C4 (two-level), C5 (where HBM-side bandwidth is the
            # bound), C5x (beyond the Infinity Cache) -- never lose the headline to an extra leg
            for leg, frames in (("c4", args.c4_frames), ("c5", args.c5_frames), ("c5x", args.c5x_frames)):
                film.clear()
                try:
                    out["roofline_" + leg] = extra_leg(pt, ctx, W, H, leg, frames, rank, live=not (args.no_live_pmc or args.pmc_child), oracle_walk=leg == "c5" and not args.no_cpu_baseline) if frames > 0 else None
                except Exception as e:
                    out["roofline_" + leg] = {"error": repr(e)}
        base = None
        if not args.no_cpu_baseline and world == 1:
            base, _, _ = cpu_baseline(arrays, scene_name, W, H, args.spp, args.depth, budget_s=args.cpu_budget_s,
                                      instances=pt.cornell_grid_instances() if args.config == "c4" else None)
            cpu_rays, cpu_spp, cpu_img = base.pop("_rays"), base.pop("_spp"), base.pop("_img")
            out["cpu_baseline"] = base
            if cpu_spp == args.spp and frame0_rays_gpu is not None:
                # the oracle rendered exactly frame 0 of this workload (the whole image): exact ray counts and every
                # float of the film must agree (checker only: none of this is in the timed region)
                out["frame0_ray_count"] = {"gpu": frame0_rays_gpu, "cpu_oracle": cpu_rays, "equal": frame0_rays_gpu == cpu_rays}
                out["frame0_film_bit_exact"] = bool(frame0_film_gpu.tobytes() == cpu_img.tobytes())
        emit(out, args)

    if presenter:
        presenter.close()
    film.close()
    scene.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
