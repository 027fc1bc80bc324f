#!/usr/bin/env python3
"""dev tool: condense a scripts/gpu_profile.sh directory into the small per-config PMC record that bench.py
reads (profiles/r01_pmc_extend*.json).  usage: make_pmc_json.py <prof dir> <out json> [config args string]
Formulas: HBM bytes = 2 x FETCH_SIZE KiB (gfx950 read correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB.
Calibrated on kernels whose byte counts are known (scripts/ubench/fetch_calib.hip under rocprofv3 --pmc,
profiles/r03_fetch_size_calibration.json, footprints 128 MB and 8 GB): streaming reads FETCH_SIZE = 0.500 x the bytes read
(the guide's x2), streaming writes WRITE_SIZE = 1.000 x the bytes written, and DIVERGENT GATHERS of 64-B records -- what the
big-scene traversal kernels do -- FETCH_SIZE = 0.97 ... 1.00 x the RECORD bytes = 0.49 ... 0.50 x the bytes of the distinct
128-B lines touched.  So 2 x FETCH_SIZE is the traffic at line granularity for every pattern (what the fabric moved), and for a
kernel that gathers 64-B records half of it is the neighbouring record nobody asked for: `hbm_read_bytes_per_ray_records64`
(1 x FETCH_SIZE) is the figure to hold against algorithmic bytes counted in 64-B records, `hbm_bytes_per_ray` (2 x FETCH_SIZE +
WRITE_SIZE) the one to hold against the 8 TB/s;
VALU issue = SQ_INSTS_VALU per second / (1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction);
wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; L2 hit = TCC_HIT / TCC_REQ."""
import json
import os
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--kernel=")]
want_kernel = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--kernel=")), None)   # e.g. --kernel=k_shade
root, out = args[0], args[1]
cfg = args[2] if len(args) > 2 else ""
s = json.load(open(os.path.join(root, "summary.json")))
bench = None
for line in open(os.path.join(root, "stats.log")):
    if line.startswith("{") and '"metric"' in line:
        bench = json.loads(line)
# the k_extend* kernel of the headline leg (the roofline_c5 leg of a default run uses another instantiation)
# dominant extend kernel of the timed region = the k_extend* kernel with most total time (the counting
# instantiation only runs the one extra untimed frame)
import re
counting = re.compile(r"k_extend<\w+, true|k_extend_inst(16)?<true|k_extend8<true")   # the instrumented instantiations (PT_FLAG_COUNT_VISITS frame)
prefix = want_kernel or "k_extend"
name = max((k for k in s["kernels"] if k.startswith(prefix) and not counting.match(k)), key=lambda k: s["kernels"][k]["total_ns"])
e, kt = s["pmc"][name], s["kernels"][name]
pl = lambda c: e.get(c + "_per_launch", 0.0)
# the bench line's block of THIS kernel (round 5: `roofline` is the kernel with the most time of the timed region, the others sit beside it)
blocks = [bench.get(k) for k in ("roofline", "roofline_extend", "roofline_shade", "roofline_wavefront") if isinstance(bench.get(k), dict)]
mine = next((b for b in blocks if str(b.get("kernel", "")).startswith(prefix.split("<")[0])), blocks[0])
# per WALKED ray: camera rays of pixels that cannot see the scene are finished without a walk (pt_stats.rays_culled) and never reach these kernels --
# bench.py's blocks carry the walked rays per launch (the fused block beside all rays, the wavefront blocks as rays_per_launch)
rays_per_launch = mine.get("rays_walked_per_launch") or mine.get("rays_per_launch") or bench["rays"] / mine["launches"]
avg_us = kt["avg_ns"] / 1e3
rec = {
    "kernel": name, "config": cfg,
    "source": f"{os.path.basename(root.rstrip('/'))}: rocprofv3 --kernel-trace --stats, then one --pmc pass per counter set "
              f"(scripts/gpu_profile.sh) on `python bench.py {cfg} --warmup 0 --no-cpu-baseline`",
    "launches": kt["calls"], "rays_per_launch_in_profile_run": rays_per_launch,
    "rocprof_avg_launch_us": avg_us,
    "bench_hipext_avg_launch_us_same_run": mine["avg_launch_us"],
    "fetch_size_kib_per_launch": pl("FETCH_SIZE"), "write_size_kib_per_launch": pl("WRITE_SIZE"),
    "hbm_read_bytes_per_launch_x2_gfx950": e["hbm_read_bytes_per_launch_gfx950_x2"],
    "hbm_write_bytes_per_launch": e["hbm_write_bytes_per_launch"], "hbm_bytes_per_launch": e["hbm_bytes_per_launch"],
    "hbm_bytes_per_ray": e["hbm_bytes_per_launch"] / rays_per_launch,
    "hbm_read_bytes_per_ray_lines128": e["hbm_read_bytes_per_launch_gfx950_x2"] / rays_per_launch,
    "hbm_read_bytes_per_ray_records64": 0.5 * e["hbm_read_bytes_per_launch_gfx950_x2"] / rays_per_launch,
    "hbm_write_bytes_per_ray": e["hbm_write_bytes_per_launch"] / rays_per_launch,
    "counter_calibration": "profiles/r03_fetch_size_calibration.json: FETCH_SIZE x2 = bytes of the 128-B lines moved (streams and gathers alike), "
                           "x1 = the 64-B records a divergent gather asked for; WRITE_SIZE x1",
    # VALU issue: wave64 VALU instructions per second of this kernel over the chip's peak of one per SIMD every 2 cycles (256 CUs x 4 SIMDs x
    # 2.4 GHz / 2: what bench.py's valu_frac uses; an fma-class op takes 4, so a kernel of fmas tops out at 0.5).  (Rounds 1-4 printed a
    # `valu_busy_fraction` = SQ_ACTIVE_INST_VALU x 4 cycles / SIMD-cycles here, which exceeded 1 for the fused kernels: the counter's unit
    # is not 4 cycles per instruction on this chip.  Dropped.)
    "valu_issue_frac": pl("SQ_INSTS_VALU") / (avg_us * 1e-6) / (256 * 4 * 2.4e9 / 2),
    "valu_wave_instr_per_64_rays": pl("SQ_INSTS_VALU") / (rays_per_launch / 64.0),
    "valu_active_lanes_per_instr": pl("SQ_THREAD_CYCLES_VALU") / pl("SQ_INSTS_VALU"),
    # the scalar unit: ONE per CU, 4.4 cycles per instruction and SIMD when all four SIMDs ask = 1 / 1.1 instructions per cycle and CU
    # (scripts/ubench/salu_rate.hip, profiles/r06z_salu_rate_ubench.txt); at the chip's nominal 2.4 GHz, like valu_issue_frac
    "salu_wave_instr_per_64_rays": pl("SQ_INSTS_SALU") / (rays_per_launch / 64.0),
    "salu_unit_frac": pl("SQ_INSTS_SALU") / (avg_us * 1e-6) / (256 * 2.4e9 / 1.1),
    "wait_any_fraction_of_wave_cycles": pl("SQ_WAIT_ANY") / pl("SQ_WAVE_CYCLES"),
    "lds_bank_conflict_fraction_of_lds_cycles": (pl("SQ_LDS_BANK_CONFLICT") / pl("SQ_LDS_IDX_ACTIVE")) if pl("SQ_LDS_IDX_ACTIVE") else None,
    "l2_hit_rate": pl("TCC_HIT_sum") / pl("TCC_REQ_sum") if pl("TCC_REQ_sum") else None,
}
rec["hbm_GBps"] = rec["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 1e9
# optional counter sets (scripts/gpu_profile.sh PMC_EXTRA=1): the vector L1's translation cache and its stalls, the L2 <-> fabric queues
for c in ("TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_PENDING_STALL_CYCLES_sum",
          "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TOTAL_CACHE_ACCESSES_sum",
          "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_LATENCY_sum",
          "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum",
          "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_TAG_STALL_sum", "TCC_BUSY_sum", "SQ_WAIT_INST_ANY", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
          "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM"):
    if c + "_per_launch" in e:
        rec.setdefault("extra_per_launch", {})[c] = pl(c)
x = rec.get("extra_per_launch", {})
if x.get("TCP_UTCL1_REQUEST_sum"):
    rec["utcl1_miss_rate"] = x.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0.0) / x["TCP_UTCL1_REQUEST_sum"]
if x.get("TCP_TCC_READ_REQ_sum") and x.get("TCP_TCC_READ_REQ_LATENCY_sum"):
    rec["l1_to_l2_read_latency_cycles"] = x["TCP_TCC_READ_REQ_LATENCY_sum"] / x["TCP_TCC_READ_REQ_sum"]
if x.get("TCP_TCC_WRITE_REQ_sum") and x.get("TCP_TCC_WRITE_REQ_LATENCY_sum"):
    rec["l1_to_l2_write_latency_cycles"] = x["TCP_TCC_WRITE_REQ_LATENCY_sum"] / x["TCP_TCC_WRITE_REQ_sum"]
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
