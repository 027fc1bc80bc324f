import re, sys
def kernels(path):
    s = open(path).read().splitlines()
    out = {}; name = None; body = []
    for l in s:
        m = re.match(r"^(_Z\w+):", l)
        if m: name = m.group(1); body = []; continue
        if l.startswith(".Lfunc_end") and name:
            ins = [x.strip().split()[0] for x in body if x.startswith("\t") and not x.strip().startswith((".", ";")) and x.strip()]
            out[name] = ins; name = None
        elif name is not None: body.append(l)
    return out
a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
flt = sys.argv[3:] 
for k in a:
    if flt and not any(f in k for f in flt): continue
    if k not in b: print("missing", k); continue
    ia, ib = a[k], b[k]
    same = ia == ib
    from collections import Counter
    ca, cb = Counter(ia), Counter(ib)
    diff = {op: cb[op] - ca[op] for op in set(ca) | set(cb) if cb[op] != ca[op]}
    print(k[:70], len(ia), "->", len(ib), "IDENTICAL opcode sequence" if same else f"differs: same multiset {not diff}; {dict(list(diff.items())[:8])}")
