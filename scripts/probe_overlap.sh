#!/bin/bash
# dev probe: does co-running two independent wavefront pipelines on one GPU raise total throughput?
run() { python bench.py --steps 32 --warmup 2 --no-cpu-baseline --no-kernel-events "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
echo -n "single: "; run
echo "two concurrent:"; run > /tmp/a.txt & run > /tmp/b.txt & wait; cat /tmp/a.txt /tmp/b.txt
echo "two concurrent, 8 frames in flight each:"; run --frames-in-flight 8 > /tmp/a.txt & run --frames-in-flight 8 > /tmp/b.txt & wait; cat /tmp/a.txt /tmp/b.txt
