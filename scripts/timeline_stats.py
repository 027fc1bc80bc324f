"""Overlap profile of the wavefront pipelines from a kernel timeline (scripts/r03_r.sh: start_us,end_us,kernel,queue,stream):
per pt_render call (ends with k_resolve) the span, per-stream busy time and gaps, and the share of time by the set of kernels
running.  usage: python scripts/timeline_stats.py timeline.csv [call index ...]"""
import collections, csv, sys
rows = [(float(r["start_us"]), float(r["end_us"]), r["kernel"], r["queue"], r["stream"]) for r in csv.DictReader(open(sys.argv[1]))]
ks = [r for r in rows if r[2] in ("extend", "shade", "generate", "resolve")]
calls, cur = [], []
for r in ks:
    cur.append(r)
    if r[2] == "resolve":
        calls.append(cur); cur = []
which = [int(a) for a in sys.argv[2:]] or range(len(calls))
for ci in which:
    c = calls[ci]
    t0 = min(r[0] for r in c); t1 = max(r[1] for r in c)
    ev = []
    for r in c:
        ev.append((r[0], 1, r[2])); ev.append((r[1], -1, r[2]))
    ev.sort()
    act = collections.Counter(); last = t0; prof = collections.Counter()
    for t, d, k in ev:
        prof[tuple(sorted(act.elements()))] += t - last; last = t
        act[k] += d
        if act[k] == 0: del act[k]
    tot = sum(prof.values())
    e = [r[1] - r[0] for r in c if r[2] == "extend"]; s = [r[1] - r[0] for r in c if r[2] == "shade"]
    top = sorted(prof.items(), key=lambda x: -x[1])[:6]
    print(f"call {ci}: {len(c)} launches, span {(t1 - t0) / 1e3:.1f} ms, extend sum {sum(e) / 1e3:.1f} ms (avg {sum(e) / len(e):.0f} us), shade sum {sum(s) / 1e3:.1f} ms (avg {sum(s) / len(s):.0f} us)")
    print("   " + "  ".join(f"{'+'.join(k) or 'idle'} {100 * v / tot:.1f}%" for k, v in top))
