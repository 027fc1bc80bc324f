cd $GRAFT_REPO_ROOT
python bench.py --config c4 --steps 8 --warmup 1 --reps 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r.get('valu_wave_instr_per_64_rays'), r.get('valu_active_lanes_per_instr'))
units=d['rays']/64
for k,b in r['valu_model']['blocks'].items():
    print('%-16s waves/unit %.3f lanes %.1f valu %d -> %.1f' % (k, b['waves']/units, b['lanes_per_wave'], b['valu'], b['waves']/units*b['valu']))
"
