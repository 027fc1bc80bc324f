"""dev model (CPU; the tree is built on the device): how many distinct 128-B LINES of 64-B nodes a ray of config C5's kind touches in the 8-wide
tree -- VERDICT r04 item 8: "half of every 128-B line is the neighbour node nobody asked for".
  as built    node i lives in line i // 2: a node's internal children are contiguous, child groups are packed back to back, so a group that starts at
              an odd index shares its first line with the previous group's last node
  aligned     every child group starts on an even index (one padding node per group with an odd start): line-mates are always SIBLINGS
  paired      ... and within a group the children are ordered so that line-mates are the pairs a ray most often visits together (slots that differ in
              ONE octant bit are neighbours across one plane; the model tries the three pairings x / y / z and the ray's own best as a bound)
Counts per ray: nodes visited, distinct lines, and the share of visited nodes whose line-mate was visited by the same ray (walk: the 7-wave kernel's,
no bound).  Cross-ray reuse in L2 (hit rate 69 % on C5x) is NOT modelled: this is the per-ray compulsory side only.
usage: sim_bvh8_lines.py [n_tris] [n_pixels]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_pix = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ctx = pt.Context(0)
v, idx, faces = pt.make_soup(n_tris, 1)
sc = pt.Scene(ctx, v, idx, faces)
nodes, prim8 = sc.read_bvh8()
V = v.reshape(-1, 3, 3).astype(np.float64)
lo_s, hi_s = V.reshape(-1, 3).min(0), V.reshape(-1, 3).max(0)
hd = nodes[:, 12:16]
o16 = np.stack([hd[:, 0] & 0xFFFF, hd[:, 0] >> 16, hd[:, 1] & 0xFFFF], 1).astype(np.float64) / 16384.0 - 2.0
ex = np.stack([(hd[:, 1] >> 16) & 31, (hd[:, 1] >> 21) & 31, hd[:, 1] >> 26], 1).astype(np.int64)
step = np.ldexp(1.0, -ex)
q = nodes[:, :12].copy().view(np.uint8).reshape(-1, 6, 8).astype(np.float64)
blo = o16[:, :, None] + q[:, 0:3, :] * step[:, :, None] - 1e-4
bhi = o16[:, :, None] + q[:, 3:6, :] * step[:, :, None] + 1e-4
child_base, imask = hd[:, 2] & 0xFFFFFF, hd[:, 2] >> 24
tri_base, lmask = hd[:, 3] & 0xFFFFFF, hd[:, 3] >> 24
c = (0.5 * (lo_s.astype(np.float32) + hi_s.astype(np.float32))).astype(np.float64)
rs = 1.0 / (0.5 * (hi_s.astype(np.float32) - lo_s.astype(np.float32))).astype(np.float64)
popc = np.array([bin(i).count("1") for i in range(256)])


def tri_hit(o, d, t_pos, tmin, tmax):
    a, b, cc = V[prim8[t_pos]]
    e1, e2 = b - a, cc - a
    p = np.cross(d, e2); det = e1 @ p
    if det == 0.0:
        return None
    tv = o - a; u = (tv @ p) / det
    qv = np.cross(tv, e1); w = (d @ qv) / det
    if u < 0 or w < 0 or u + w > 1:
        return None
    t = (e2 @ qv) / det
    return t if tmin < t < tmax else None


def trace(o, d, tmin=1e-3, tmax=1e4):
    """the walk without a pending-group bound -> (t, triangle position, [(parent, slot)] of every node visited below the root)"""
    on, dn = (o - c) * rs, d * rs
    inv = 1.0 / np.where(dn == 0.0, 1e-300, dn)
    octant = int(inv[0] < 0) | (int(inv[1] < 0) << 1) | (int(inv[2] < 0) << 2)
    best_t, best = tmax, -1
    seen = []
    stack = []
    cur = 0
    while True:
        n = cur
        t0 = (blo[n] - on[:, None]) * inv[:, None]; t1 = (bhi[n] - on[:, None]) * inv[:, None]
        tn = np.maximum(np.minimum(t0, t1).max(0), tmin); tf = np.minimum(np.maximum(t0, t1).min(0), best_t)
        hit = (tn <= tf) & ((((int(imask[n]) | int(lmask[n])) >> np.arange(8)) & 1) > 0)
        for s in range(8):
            if hit[s] and (int(lmask[n]) >> s) & 1:
                pos = int(tri_base[n]) + int(popc[int(lmask[n]) & ((1 << s) - 1)])
                t = tri_hit(o, d, pos, tmin, tmax)
                if t is not None and t < best_t:
                    best_t, best = t, pos
        kids = sorted((s for s in range(8) if hit[s] and (int(imask[n]) >> s) & 1), key=lambda s: s ^ octant)
        for s in reversed(kids):
            stack.append((n, s))
        if not stack:
            return best_t, best, seen
        n, s = stack.pop()
        seen.append((n, s))
        cur = int(child_base[n]) + int(popc[int(imask[n]) & ((1 << s) - 1)])


rng = np.random.default_rng(5)
W, H = 1920, 1080
tot = dict(rays=0, nodes=0, lines_built=0, mates_built=0, lines_aligned=0, mates_aligned=0, lines_x=0, lines_y=0, lines_z=0, lines_best=0)
t_start = time.time()
for _ in range(n_pix):
    px, py = rng.integers(0, W), rng.integers(0, H)
    sx, sy = (px + rng.random()) / W * 2 - 1, (py + rng.random()) / H * 2 - 1
    o = np.array([0.0, -1.0, 5.0]); tgt = np.array([sx, sy - 1.0, 2.0])
    d = tgt - o; d /= np.linalg.norm(d)
    for depth in range(16):
        t, pos, seen = trace(o, d)
        tot["rays"] += 1
        tot["nodes"] += len(seen) + 1
        idxs = {int(child_base[n]) + int(popc[int(imask[n]) & ((1 << s) - 1)]) for n, s in seen} | {0}
        lines = {i // 2 for i in idxs}
        tot["lines_built"] += len(lines)
        tot["mates_built"] += sum(1 for i in idxs if (i ^ 1) in idxs)
        # aligned groups: line = (parent, rank within the group // 2)
        ranks = {(n, int(popc[int(imask[n]) & ((1 << s) - 1)])) for n, s in seen}
        la = {(n, r // 2) for n, r in ranks}
        tot["lines_aligned"] += len(la) + 1
        tot["mates_aligned"] += sum(1 for n, r in ranks if (n, r ^ 1) in ranks)
        # paired by one octant bit (needs eight slots per group in memory, i.e. empty slots kept: an upper bound on what pairing can give)
        per = {}
        for axis, key in ((1, "lines_x"), (2, "lines_y"), (4, "lines_z")):
            per[key] = len({(n, min(s, s ^ axis)) for n, s in seen}) + 1
            tot[key] += per[key]
        tot["lines_best"] += min(per.values())
        if pos < 0:
            break
        a, b, cc = V[prim8[pos]]
        nrm = -np.cross(b - a, cc - a); nrm /= np.linalg.norm(nrm)
        o = o + t * d
        while True:
            w = rng.normal(size=3); w /= np.linalg.norm(w)
            if w @ nrm > 0:
                break
        d = w
r = tot["rays"]
print(f"{n_tris} triangles, {len(nodes)} nodes, {n_pix} paths, {r} rays, {time.time() - t_start:.1f} s")
print(f"nodes visited per ray {tot['nodes'] / r:.2f}")
print(f"as built : distinct 128-B lines per ray {tot['lines_built'] / r:.2f} ({128 * tot['lines_built'] / r:.0f} B), visited nodes whose line-mate was visited too {100 * tot['mates_built'] / tot['nodes']:.1f} %")
print(f"aligned  : {tot['lines_aligned'] / r:.2f} lines ({128 * tot['lines_aligned'] / r:.0f} B), {100 * tot['mates_aligned'] / tot['nodes']:.1f} %  -> {100 * (tot['lines_aligned'] / tot['lines_built'] - 1):+.1f} % lines")
print(f"paired by one octant bit, all eight slots kept in memory (x / y / z / the ray's best of the three): {tot['lines_x'] / r:.2f} / {tot['lines_y'] / r:.2f} / {tot['lines_z'] / r:.2f} / {tot['lines_best'] / r:.2f} lines")
