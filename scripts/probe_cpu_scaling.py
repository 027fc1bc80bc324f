#!/usr/bin/env python3
"""dev tool (GPU box): how many host cores does this container really get?  nproc / affinity / cgroup quota, and the
oracle's Mrays/s at 1 .. all threads on a 1080p 1-spp Cornell frame (the cpu_baseline workload of bench.py)."""
import importlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
from oracle import pt_oracle as orc
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.strerror)
print("loadavg", open("/proc/loadavg").read().strip())
osc = orc.Scene(*pt.load_obj(pt.ASSET_CORNELL))
p = orc.default_params(width=1920, height=1080, spp_per_frame=1, max_depth=8)
for nt in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1):
        break
    t0 = time.perf_counter(); _, rays, _, _ = osc.render_frame(p, mode=1, nthreads=nt); dt = time.perf_counter() - t0
    t0c = time.process_time(); osc.render_frame(p, mode=1, nthreads=nt); cpu = time.process_time() - t0c
    print(f"threads {nt:4d}: {rays / dt / 1e6:8.2f} Mrays/s   wall {dt:6.3f} s   cpu-seconds of a second run {cpu:7.3f}")
