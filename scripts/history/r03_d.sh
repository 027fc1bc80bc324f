#!/bin/bash
# round 3, session d: PLOC with the parity tie-break and the keep-the-cheaper rule -- probe at radius 8 / 16 / 32, GPU tests
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for r in 8 16 32; do
  PT_TUNE=ploc_radius=$r timeout 900 python scripts/probe_stress_scene.py > $O/r03d_probe_stress_scene_r$r.txt 2>&1; echo "probe r=$r rc=$?"; tail -5 $O/r03d_probe_stress_scene_r$r.txt
done
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r03d_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r03d_pytest.txt
