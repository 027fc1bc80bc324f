one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print(d['value'], 'ms', d['ms_per_step'], 'rounds', d['rounds'], 'ext', r.get('extend_ms'), 'sh', r.get('shade_ms'), 'launches', r.get('launches'))"; }
for s in 64 256 512; do echo -n "${s}x${s} K1: "; one --width $s --height $s --steps 1 --frames-in-flight 1 --sample-groups 1; echo -n "${s}x${s} K1 noevents: "; one --width $s --height $s --steps 1 --frames-in-flight 1 --sample-groups 1 --no-kernel-events; done
echo -n "1080p K16 noevents: "; one --no-kernel-events
echo -n "1080p K16 events: "; one
