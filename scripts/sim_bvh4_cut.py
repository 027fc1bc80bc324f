"""dev tool (round 6): the BVH4 cut of the Cornell box offline -- the binary surface-area sweep of bvh4_sah_device.hip restated, its greedy cut (rounds 1-5; reproduces the
device tree: 8 nodes) and the least-area cut by dynamic programming, and a simulation of the nearest-first walk over path-traced rays counting node and leaf visits per
ray under both (3.42 / 1.39 -> 2.87 / 1.40; the device counters: 3.41 / 1.37 -> 2.88 / 1.36).  usage: python scripts/sim_bvh4_cut.py [paths]"""
import sys, os, numpy as np, importlib, itertools
sys.path.insert(0,'/root/repo')
pt = importlib.import_module("single-file-vulkan-pathtracing_amd")
v,i,f = pt.load_obj(pt.ASSET_CORNELL)
V = np.asarray(v,dtype=np.float64).reshape(-1,3); I=np.asarray(i).reshape(-1,3); F=np.asarray(f,dtype=np.float64).reshape(-1,6)
tris = V[I]            # (36,3,3)
nq = len(tris)//2
quads = [(2*q, 2*q+1) for q in range(nq)]
qlo = np.array([np.minimum(tris[a].min(0), tris[b].min(0)) for a,b in quads]); qhi = np.array([np.maximum(tris[a].max(0), tris[b].max(0)) for a,b in quads])
def area(lo,hi):
    d=np.maximum(hi-lo,0); return 2*(d[0]*d[1]+d[1]*d[2]+d[2]*d[0])
# ---- binary SAH sweep (bvh4_sah_device.hip: cost area(L) n(L) + area(R) n(R), first minimum in (cost, axis, position) order)
class N: pass
def build(ids):
    n=N(); n.ids=ids; n.lo=qlo[ids].min(0); n.hi=qhi[ids].max(0); n.l=n.r=None
    if len(ids)==1: return n
    best=None
    cen=(qlo[ids]+qhi[ids])*0.5
    for ax in range(3):
        order=sorted(range(len(ids)), key=lambda k:(cen[k,ax], ids[k]))
        for pos in range(len(ids)-1):
            L=[ids[k] for k in order[:pos+1]]; R=[ids[k] for k in order[pos+1:]]
            c=area(qlo[L].min(0),qhi[L].max(0))*len(L)+area(qlo[R].min(0),qhi[R].max(0))*len(R)
            key=(c,ax,pos)
            if best is None or key<best[0]: best=(key,L,R)
    n.l=build(best[1]); n.r=build(best[2]); return n
root=build(list(range(nq)))
# ---- collapse to BVH4: greedy (open the internal child of largest area) and DP-optimal (min sum of internal-node areas)
class W: pass
def greedy(n):
    w=W(); w.lo=n.lo; w.hi=n.hi; ch=[n.l,n.r]
    while len(ch)<4:
        cand=[c for c in ch if c.l is not None]
        if not cand: break
        o=max(cand,key=lambda c:area(c.lo,c.hi)); k=ch.index(o); ch[k:k+1]=[o.l,o.r]
    w.ch=[greedy(c) if c.l is not None else c for c in ch]; return w
import functools
memo={}
def cost_forest(n,k):   # min (sum of internal wide-node areas, structure) to cover subtree n with <= k wide children
    key=(id(n),k)
    if key in memo: return memo[key]
    if n.l is None: r=(0.0,[n])
    else:
        best=None
        if k>=1:
            c,chs=cost_forest_split(n,4); best=(area(n.lo,n.hi)+c,[('node',n,chs)])
        if k>=2:
            c,chs=cost_forest_split(n,k)
            if c<best[0]-1e-12: best=(c,chs)
        r=best
    memo[key]=r; return r
def cost_forest_split(n,k):
    best=None
    for j in range(1,k):
        cl,sl=cost_forest(n.l,j); cr,sr=cost_forest(n.r,k-j)
        if best is None or cl+cr<best[0]-1e-12: best=(cl+cr,sl+sr)
    return best
def dp(n):
    c,chs=cost_forest_split(n,4)
    def mk(lo,hi,chs):
        w=W(); w.lo=lo; w.hi=hi; w.ch=[]
        for c in chs:
            if isinstance(c,tuple): w.ch.append(mk(c[1].lo,c[1].hi,c[2]))
            else: w.ch.append(c)
        return w
    return mk(n.lo,n.hi,chs), area(n.lo,n.hi)+c
def stats(w,ra):
    s=area(w.lo,w.hi)/ra; cnt=1
    for c in w.ch:
        if isinstance(c,W): a,b=stats(c,ra); s+=a; cnt+=b
    return s,cnt
ra=area(root.lo,root.hi)
g=greedy(root); d,dc=dp(root)
print("greedy: SAH node visits %.3f nodes %d"%stats(g,ra)); print("dp:     SAH node visits %.3f nodes %d"%stats(d,ra))
# ---- traversal simulator
rng=np.random.default_rng(1)
def ray_tri(o,dr,t3):
    e1=t3[1]-t3[0]; e2=t3[2]-t3[0]; p=np.cross(dr,e2); det=e1@p
    if abs(det)<1e-14: return None
    s=o-t3[0]; u=(s@p)/det
    if u<0 or u>1: return None
    q=np.cross(s,e1); vv=(dr@q)/det
    if vv<0 or u+vv>1: return None
    t=(e2@q)/det
    return t if t>1e-3 else None
def box_t(o,inv,lo,hi,tmax):
    t0=(lo-o)*inv; t1=(hi-o)*inv
    tn=max(np.minimum(t0,t1).max(),1e-3); tf=min(np.maximum(t0,t1).min(),tmax)
    return tn if tn<=tf else None
def trace(w,o,dr,count):
    with np.errstate(divide='ignore'): inv=1.0/dr
    best=(1e4,None); stack=[(0.0,w)]
    while stack:
        tn,n=stack.pop()
        if tn>best[0]: continue
        if isinstance(n,W):
            count[0]+=1
            hits=[]
            for c in n.ch:
                t=box_t(o,inv,c.lo,c.hi,best[0])
                if t is not None: hits.append((t,c))
            hits.sort(key=lambda x:-x[0])
            stack.extend(hits)
        else:
            count[1]+=1
            q=n.ids[0]
            for tix in quads[q]:
                t=ray_tri(o,dr,tris[tix])
                if t is not None and t<best[0]: best=(t,tix)
    return best
def normal(tix):
    t3=tris[tix]; n=np.cross(t3[1]-t3[0],t3[2]-t3[0]); return -n/np.linalg.norm(n)
def sample_paths(npaths):
    rays=[]
    for _ in range(npaths):
        # camera ray inside the box's projection (the culled border is not walked)
        px=rng.uniform(0.125,0.875); py=rng.uniform(0.125,0.875)
        d=np.array([px*2-1, py*2-1 -1+1, 0.0]); tgt=np.array([d[0], (py*2-1)-1.0, 2.0]); o=np.array([0.0,-1.0,5.0]); dr=tgt-o; dr/=np.linalg.norm(dr)
        for depth in range(8):
            rays.append((o,dr))
            t,tix=trace(g,o,dr,[0,0])
            if tix is None: break
            p=o+t*dr; n=normal(tix)
            # uniform hemisphere about n (the reference does not flip normals: directions may point into the surface)
            r1=rng.uniform(); r2=rng.uniform(); sq=np.sqrt(1-r1*r1); phi=2*np.pi*r2
            a=np.array([1,0,0]) if abs(n[0])<0.9 else np.array([0,1,0]); T=np.cross(n,a); T/=np.linalg.norm(T); B=np.cross(n,T)
            dr=T*np.cos(phi)*sq+B*np.sin(phi)*sq+n*r1; o=p
    return rays
rays=sample_paths(int(sys.argv[1]) if len(sys.argv)>1 else 1500)
print("rays",len(rays))
for name,w in (("greedy",g),("dp",d)):
    c=[0,0]
    for o,dr in rays: trace(w,o,dr,c)
    print(name,"node visits/ray %.3f leaf visits/ray %.3f"%(c[0]/len(rays),c[1]/len(rays)))
def show(w,ind=0):
    print(" "*ind+"node area %.2f"%(area(w.lo,w.hi)/ra), [("N" if isinstance(c,W) else "q%d(%.2f)"%(c.ids[0],area(c.lo,c.hi)/ra)) for c in w.ch])
    for c in w.ch:
        if isinstance(c,W): show(c,ind+2)
print("greedy tree:"); show(g); print("dp tree:"); show(d)
