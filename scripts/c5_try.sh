#!/bin/bash
# C5 (BASELINE.json config 5): 1M-triangle soup, 1920x1080, 16 spp, depth 16
set -e
mkdir -p /tmp/c5 gpurun_out
python - <<'PY'
import importlib, time
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
t=time.time(); pt.write_soup_obj("/tmp/c5/soup1m.obj", 1000000, 1); print("soup written in %.2fs" % (time.time()-t))
PY
ls -la /tmp/c5
./single-file-vulkan-pathtracing_amd/pt_main --obj /tmp/c5/soup1m.obj --width 1920 --height 1080 --frames 1 --spp 16 --depth 16 --ppm gpurun_out/c5.ppm
./single-file-vulkan-pathtracing_amd/pt_main --obj /tmp/c5/soup1m.obj --width 1920 --height 1080 --frames 2 --spp 16 --depth 16
