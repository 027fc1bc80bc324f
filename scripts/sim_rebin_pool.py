"""Model of block-/wave-level re-binning: a pool of S ray slots per wave64, each iteration runs the phase
(node step / leaf step / finish+refill) with the most rays for up to 64 of them.  Phase transitions are a Markov chain calibrated on the
Cornell kernel's device counters (3.11 node steps, 1.26 leaf steps per ray); `ovh` = VALU instructions of state movement and list
building per step.  Prints VALU wave-instructions per 64 rays and lanes per instruction by pool size."""
import numpy as np, sys
rng = np.random.default_rng(1)
P = {0: [(0.55, 0), (0.30, 1), (0.15, 2)],   # node -> node / leaf / done
     1: [(0.30, 0), (0.30, 1), (0.40, 2)]}   # leaf -> node / leaf / done
def sim(S, nrays=200000, ovh=(12, 15, 15), policy="max", lanes=64):
    # phases: 0 node, 1 leaf, 2 done(= needs finish+refill)
    ph = np.full(S, 2)
    issued = 0; valu = 0; lane_instr = 0; execs = [0,0,0]; lanesum=[0,0,0]
    cost = {0: 90 + ovh[0], 1: 80 + 39 + ovh[1], 2: 113 + 5 + ovh[2]}
    done_rays = 0
    while done_rays < nrays:
        cnt = [np.sum(ph == k) for k in range(3)]
        if issued >= nrays: cnt[2] = 0
        if policy == "max":
            k = int(np.argmax(cnt))
        if cnt[k] == 0: break
        idx = np.nonzero(ph == k)[0][:lanes]
        n = len(idx)
        execs[k] += 1; lanesum[k] += n
        valu += cost[k]; lane_instr += cost[k] * n
        if k == 2:
            issued += n; ph[idx] = 0
        else:
            r = rng.random(n)
            p = P[k]
            nxt = np.where(r < p[0][0], p[0][1], np.where(r < p[0][0] + p[1][0], p[1][1], 2))
            done_rays += int(np.sum(nxt == 2))
            ph[idx] = nxt
    return valu / nrays * 64, lane_instr / valu, [lanesum[k] / max(1, execs[k]) for k in range(3)], [e / nrays * 64 for e in execs]
for S in (64, 80, 96, 128, 192, 256):
    v, l, ls, ex = sim(S)
    print(f"S={S:4d}: VALU/64 rays {v:7.1f}  lanes/instr {l:5.1f}  lanes node/leaf/refill {ls[0]:.1f} {ls[1]:.1f} {ls[2]:.1f}  execs {ex[0]:.2f} {ex[1]:.2f} {ex[2]:.2f}")
