"""dev model (CPU, on a box with a GPU only because the 8-wide tree is built on the device): node visits per ray of the byte-plane tree under
several pending-children policies, on paths of config C5's kind (camera rays + uniform-hemisphere bounces through the 1 M-triangle soup).
  kernel    what k_extend8 does: children in octant order, the rest of a node pending as ONE entry culled by the smallest entry distance of the node's hits
  all8      ... of ALL eight slots' entry distances, hit or not (no select per child in the kernel)
  none      no bound at all: every pending child is visited
  int_min   ... of the node's INTERNAL hits (its leaf triangles were tested at the visit and are not pending)
  rest_fix  ... of the internal hits except the one visited first, fixed when the entry is made
  m12       ... the cheap form of rest_fix: the two smallest entry distances m1 <= m2 of ALL the node's hits (leaves included) and whose m1 is:
            m2 if the child visited first is the nearest hit, else m1
  rest_min  ... culled by the smallest entry distance of the children still pending (not of all the node's hits)
  per_child every pending child remembers its own entry distance and is skipped when the best hit got closer meanwhile
  sorted    per_child + children of a node visited nearest first
  global    one priority queue over all pending children (a lower bound for this tree)
usage: sim_bvh8_policies.py [n_tris] [n_pixels]"""
import heapq, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_pix = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ctx = pt.Context(0)
v, idx, faces = pt.make_soup(n_tris, 1)
sc = pt.Scene(ctx, v, idx, faces)
nodes, prim8 = sc.read_bvh8()
info = sc.info()
V = v.reshape(-1, 3, 3).astype(np.float64)          # by primitive id
lo_s, hi_s = V.reshape(-1, 3).min(0), V.reshape(-1, 3).max(0)
# the normalised scene box of the tree (k_w8_emit: (x - c) * rs): recovered from the root's decoded box against the scene's box
hd = nodes[:, 12:16]
o16 = np.stack([hd[:, 0] & 0xFFFF, hd[:, 0] >> 16, hd[:, 1] & 0xFFFF], 1).astype(np.float64) / 16384.0 - 2.0
ex = np.stack([(hd[:, 1] >> 16) & 31, (hd[:, 1] >> 21) & 31, hd[:, 1] >> 26], 1).astype(np.int64)
step = np.ldexp(1.0, -ex)
q = nodes[:, :12].copy().view(np.uint8).reshape(-1, 6, 8).astype(np.float64)      # rows lo.x lo.y lo.z hi.x hi.y hi.z, 8 children each
blo = o16[:, :, None] + q[:, 0:3, :] * step[:, :, None]
bhi = o16[:, :, None] + q[:, 3:6, :] * step[:, :, None]
child_base, imask = hd[:, 2] & 0xFFFFFF, hd[:, 2] >> 24
tri_base, lmask = hd[:, 3] & 0xFFFFFF, hd[:, 3] >> 24
# the tree's normalised coordinates (lbvh_build.hip ptb_norm_box): centre and half extent of the scene's bounds; the decoded boxes get a margin
# of 1e-4 here for what that guess may be off by (the tree's own boxes are padded by 4e-6)
c = (0.5 * (lo_s.astype(np.float32) + hi_s.astype(np.float32))).astype(np.float64)
rs = 1.0 / (0.5 * (hi_s.astype(np.float32) - lo_s.astype(np.float32))).astype(np.float64)
blo -= 1e-4; bhi += 1e-4
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


def trace(o, d, policy, tmin=1e-3, tmax=1e4):
    """-> (t, triangle position, node visits, triangle tests)"""
    on, dn = (o - c) * rs, d * rs
    inv = 1.0 / np.where(dn == 0.0, 1e-300, dn)
    octant = int(inv[0] < 0) | (int(inv[1] < 0) << 1) | (int(inv[2] < 0) << 2)
    best_t, best = tmax, -1
    visits = tests = 0
    stack = []          # kernel / rest_min: [base, imask, pending priority list [(slot, tn)], gmin]; per_child / sorted: same with per-child culling
    heap = []           # global: (tn, node)
    cur = 0
    while True:
        visits += 1
        n = cur
        t0 = (blo[n] - on[:, None]) * inv[:, None]; t1 = (bhi[n] - on[:, None]) * inv[:, None]
        tn = np.maximum(np.minimum(t0, t1).max(0), tmin); tf = np.minimum(np.maximum(t0, t1).min(0), best_t)
        hit = (tn <= tf) & ((((int(imask[n]) | int(lmask[n])) >> np.arange(8)) & 1) > 0)
        # the node's leaf triangles first
        for s in range(8):
            if hit[s] and (int(lmask[n]) >> s) & 1:
                tests += 1
                pos = int(tri_base[n]) + int(popc[int(lmask[n]) & ((1 << s) - 1)])
                t = tri_hit(o, d, pos, tmin, tmax)
                if t is not None and t < best_t:
                    best_t, best = t, pos
        kids = [(s, float(tn[s])) for s in range(8) if hit[s] and (int(imask[n]) >> s) & 1]
        gmin_all = float(tn[hit].min()) if hit.any() else np.inf
        if policy == "global":
            for s, t in kids:
                heapq.heappush(heap, (t, int(child_base[n]) + int(popc[int(imask[n]) & ((1 << s) - 1)])))
            cur = -1
            while heap:
                t, nd = heapq.heappop(heap)
                if t <= best_t:
                    cur = nd
                    break
            if cur < 0:
                return best_t, best, visits, tests
            continue
        if policy == "sorted":
            kids.sort(key=lambda k: k[1])
        else:
            kids.sort(key=lambda k: k[0] ^ octant)
        cur = -1
        if kids:   # the first child in the policy's order is visited right away; the rest waits as one entry
            s0, _ = kids.pop(0)
            cur = int(child_base[n]) + int(popc[int(imask[n]) & ((1 << s0) - 1)])
            if kids:
                g = gmin_all
                if policy == "int_min":
                    g = min(min(k[1] for k in kids), _)
                if policy == "all8":
                    g = float(tn[((int(imask[n]) | int(lmask[n])) >> np.arange(8)) & 1 > 0].min())
                if policy == "none":
                    g = -np.inf
                if policy == "rest_fix":
                    g = min(k[1] for k in kids)
                if policy == "m12":
                    allh = sorted((float(tn[s]), s) for s in range(8) if hit[s])
                    g = (allh[1][0] if len(allh) > 1 else np.inf) if allh[0][1] == s0 else allh[0][0]
                stack.append([n, kids, g])
        while cur < 0 and stack:
            nd, ks, gmin = stack[-1]
            if policy == "kernel" and gmin > best_t:
                stack.pop(); continue
            if policy == "rest_min" and min(k[1] for k in ks) > best_t:
                stack.pop(); continue
            if policy in ("int_min", "rest_fix", "m12", "all8", "none") and gmin > best_t:
                stack.pop(); continue
            s, t = ks.pop(0)
            if not ks:
                stack.pop()
            if policy in ("per_child", "sorted") and t > best_t:
                continue
            cur = int(child_base[nd]) + int(popc[int(imask[nd]) & ((1 << s) - 1)])
        if cur < 0:
            return best_t, best, visits, tests


rng = np.random.default_rng(5)
W, H = 1920, 1080
policies = ["kernel", "all8", "none", "int_min", "rest_fix", "m12", "rest_min", "per_child", "sorted", "global"]
tot = {p: [0, 0] for p in policies}
n_rays = 0
t_start = time.time()
for _ in range(n_pix):
    px, py = rng.integers(0, W), rng.integers(0, H)
    sx, sy = (px + rng.random()) / W * 2 - 1, (py + rng.random()) / H * 2 - 1
    o = np.array([0.0, -1.0, 5.0]); tgt = np.array([sx, sy - 1.0, 2.0])
    d = tgt - o; d /= np.linalg.norm(d)
    for depth in range(16):
        res = {p: trace(o, d, p) for p in policies}
        ts = {round(r[0], 9) for r in res.values()}
        assert len(ts) == 1, ("policies disagree on the hit", res)
        for p in policies:
            tot[p][0] += res[p][2]; tot[p][1] += res[p][3]
        n_rays += 1
        t, pos = res["kernel"][0], res["kernel"][1]
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
print(f"{n_tris} triangles, {info.n_wide8_nodes} nodes, {n_pix} paths, {n_rays} rays ({n_rays / n_pix:.2f} per path), {time.time() - t_start:.1f} s")
for p in policies:
    print(f"{p:10s} node visits per ray {tot[p][0] / n_rays:7.2f}   triangle tests per ray {tot[p][1] / n_rays:5.2f}")
