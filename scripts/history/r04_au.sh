#!/bin/bash
# round 4, GPU session AU: last check of the final tree: suite, smoke, pt_main with both pipelines (host loader + image writers), the loader on C5's OBJ
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04au_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/r04au_pytest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
P=single-file-vulkan-pathtracing_amd/pt_main
$P --width 640 --height 360 --frames 3 --pfm /tmp/wf.pfm > $O/r04au_pt_main_wavefront.txt 2>&1; $P --width 640 --height 360 --frames 3 --pipeline fused --pfm /tmp/fu.pfm > $O/r04au_pt_main_fused.txt 2>&1
cmp /tmp/wf.pfm /tmp/fu.pfm && echo "pt_main: fused image == wavefront image"; tail -2 $O/r04au_pt_main_fused.txt
timeout 600 python bench.py --config c5 --steps 4 --no-cpu-baseline --no-extra-legs --no-live-pmc 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c5', d['value'], d['ms_per_step'], d.get('ingest'), d['bvh']['build_ms'])"
