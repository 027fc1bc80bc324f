"""The same model with R lane-private rays (no cross-lane movement: a lane works on whichever of ITS rays is in the wave's phase)."""
import numpy as np, sys
rng = np.random.default_rng(1)
P = {0: (0.55, 0.30), 1: (0.30, 0.30)}   # node -> node / leaf / (done); leaf -> node / leaf / (done)
def sim(R, nrays=400000, ovh=20, policy="max", hitfrac=0.9):
    L = 64
    ph = np.full((R, L), 2)          # 0 node 1 leaf 2 done/needs refill 3 exhausted
    issued = 0; valu = 0.0; lane_instr = 0.0; execs = [0,0,0]; lanesum=[0,0,0]
    cost = {0: 90 + 8 + ovh, 1: 80 + 39 + 4 + ovh, 2: 113 + 5 + ovh}   # + pop iterations
    done_rays = 0
    while True:
        has = [(ph == k).any(axis=0) for k in range(3)]
        cnt = [int(h.sum()) for h in has]
        if issued >= nrays: cnt[2] = 0
        k = int(np.argmax(cnt))
        if cnt[k] == 0: break
        lanes = np.nonzero(has[k])[0]
        j = np.argmax(ph[:, lanes] == k, axis=0)
        n = len(lanes)
        execs[k] += 1; lanesum[k] += n
        valu += cost[k]; lane_instr += cost[k] * n
        if k == 2:
            issued += n; ph[j, lanes] = 0
        else:
            r = rng.random(n); p = P[k]
            ph[j, lanes] = np.where(r < p[0], 0, np.where(r < p[0] + p[1], 1, 2))
    return valu / issued * 64, lane_instr / valu, [lanesum[k] / max(1, execs[k]) for k in range(3)], [e / issued * 64 for e in execs]
for R in (1, 2, 3, 4):
    for ovh in (0, 20):
        v, l, ls, ex = sim(R, ovh=ovh)
        print(f"R={R} ovh={ovh:2d}: VALU/64 rays {v:7.1f}  lanes/instr {l:5.1f}  lanes node/leaf/refill {ls[0]:.1f} {ls[1]:.1f} {ls[2]:.1f}  execs {ex[0]:.2f} {ex[1]:.2f} {ex[2]:.2f}")
