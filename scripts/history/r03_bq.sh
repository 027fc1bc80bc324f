#!/bin/bash
# round 3, session bq: do the allocation levels of the C2 frame rate follow how the workspace is mapped?  hipMalloc against the
# virtual-memory API (one physical handle per array) at 2 MB / 1 GB / 4 GB virtual alignment; 4 allocations per process, 3 processes each, interleaved
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2 3; do
  for m in 0 2 1024 4096; do
    echo -n "ws_align_mb=$m: "; PT_TUNE="ws_align_mb=$m" timeout 300 python scripts/probe_alloc_modes.py 4 2>&1 | tail -1
  done
done 2>&1 | tee $O/r03bq_alloc_vmm.log
