#!/bin/bash
# round 4, GPU session AC: the multi-threaded OBJ loader on the GPU box's host (1 M-triangle soup, 139 MB of text)
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY' | tee $O/r04ac_load_obj.log
import importlib, time, os, hashlib
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
t0 = time.perf_counter(); pt.write_soup_obj("/tmp/soup1m.obj", 1000000, 1); print("write_s", round(time.perf_counter() - t0, 3), os.path.getsize("/tmp/soup1m.obj"))
for k in range(4):
    t1 = time.perf_counter(); a = pt.load_obj("/tmp/soup1m.obj"); print("load_s", round(time.perf_counter() - t1, 3))
b = pt.make_soup(1000000, 1)
print("equals make_soup:", all(x.tobytes() == y.tobytes() for x, y in zip(a, b)))
PY
