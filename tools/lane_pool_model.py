"""Would more rays than lanes per wave pay?  ("lanes that do work" -- k_trace_w4 runs ~61 % of its lanes per pass.)

tools/wave_schedule_model.py replays the kernel's wave-level schedule (phase A retire / refill, loop B leaves, loop C wide nodes) over the step
sequences of real rays and counts passes; here the same replay prices three ways out in VECTOR INSTRUCTIONS PER RAY (static per-pass counts of
the three loops, tools/isa_mix.py --loops: A 220, B 140, C 179), the quantity the kernel is bound by (0.895 of the VALU issue ceiling):

  greedy       run whichever loop serves the most lanes per instruction issued -- is the shipped 32:8 rule leaving anything on the table?
  quorum       phase A waits until >= q lanes are idle (the kernel refills as soon as ONE is, 12 rays per 220-instruction pass)
  R rays/lane  every lane owns R rays, the current one in registers, the others parked in LDS (17 dwords each + a traversal stack each);
               a pass serves a lane if ANY of its rays is in the wanted state, switching first (SW instructions, paid by the whole wave in
               every pass in which at least `swmin` lanes switch)

No GPU.  usage: python tools/lane_pool_model.py [--rays-per-queue 4000] [--width 160 --height 90]
Result (profiles/r05_lane_pool_model.log): every variant lands within -7 % .. +20 % of the shipped schedule's 110 instructions per closest-hit
ray; two rays per lane at the best setting saves 5 - 7 % of the instructions and needs 2 x the LDS per wave (stacks + 4.3 KB of parked state:
13 instead of 26 waves per CU, where the bare visit chain runs at 0.83 x the rate, profiles/r04_visit_microbench.json) -- a loss.  A pool with
FREE assignment of rays to lanes (every pass full) would need the whole ray state in LDS for 2 x 64 rays (21 KB per wave with the stacks: 7
waves per CU) and is capped by the L1 / texture-address path (0.75 busy: x 1.19) before the vector ALU's 0.895 stops binding."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import wave_schedule_model as M

N, L, TT = M.N, M.L, M.TT
COST = dict(A=220, B=140, C=179)


def _kind(ev, ln, stride, ray, pos):
    k = np.zeros(ray.shape, np.int64)
    idx = np.nonzero(ray >= 0)
    if len(idx[0]):
        k[idx] = ev[ray[idx], np.minimum(pos[idx], stride - 1)]
        done = pos[idx] >= ln[ray[idx]]
        k[tuple(i[done] for i in idx)] = 0
    return k


def replay_greedy(ev, ln):
    n = len(ln); stride = ev.shape[1]; ln = np.minimum(ln, stride)
    ray = np.full(64, -1, np.int64); pos = np.zeros(64, np.int64); nxt = 0
    P = dict(A=0, B=0, C=0, S=0); Ln = dict(A=0, B=0, C=0, S=0)
    while True:
        k = _kind(ev, ln, stride, ray, pos)
        nA = int((k == 0).sum()) if nxt < n else 0
        nB = int(((k == L) | (k == TT)).sum()); nC = int((k == N).sum())
        if nA == 0 and nB == 0 and nC == 0:
            break
        sA, sB, sC = nA / COST["A"], nB / COST["B"], nC / COST["C"]
        if sA >= sB and sA >= sC and nA:
            need = np.where(k == 0)[0]; take = min(len(need), n - nxt)
            P["A"] += 1; Ln["A"] += len(need)
            ray[need] = -1; ray[need[:take]] = np.arange(nxt, nxt + take); pos[need[:take]] = 0; nxt += take
        elif sB >= sC and nB:
            leaf = (k == L) | (k == TT); P["B"] += 1; Ln["B"] += nB; pos[leaf] += 1
        else:
            node = k == N; P["C"] += 1; Ln["C"] += nC; pos[node] += 1
    return P, Ln


def replay_quorum(ev, ln, node_q, leaf_q, rq):
    n = len(ln); stride = ev.shape[1]; ln = np.minimum(ln, stride)
    ray = np.full(64, -1, np.int64); pos = np.zeros(64, np.int64); nxt = 0
    P = dict(A=0, B=0, C=0, S=0); Ln = dict(A=0, B=0, C=0, S=0)
    while True:
        k = _kind(ev, ln, stride, ray, pos)
        idle = k == 0
        busy = int((k != 0).sum())
        if idle.any() and nxt < n and (int(idle.sum()) >= rq or busy < 16):
            need = np.where(idle)[0]; take = min(len(need), n - nxt)
            P["A"] += 1; Ln["A"] += len(need)
            ray[need] = -1; ray[need[:take]] = np.arange(nxt, nxt + take); pos[need[:take]] = 0; nxt += take
            k = _kind(ev, ln, stride, ray, pos)
        exhausted = nxt >= n
        if not (k != 0).any():
            if exhausted:
                break
            continue
        leaf = (k == L) | (k == TT); n_node = int((k == N).sum())
        if leaf.any() and (int(leaf.sum()) >= leaf_q or n_node < node_q):
            while True:
                P["B"] += 1; Ln["B"] += int(leaf.sum()); pos[leaf] += 1
                k = _kind(ev, ln, stride, ray, pos); leaf = (k == L) | (k == TT)
                if not (leaf.any() and int(leaf.sum()) >= leaf_q):
                    break
        while True:
            node = k == N
            if not node.any():
                break
            if int(node.sum()) < node_q:
                if ((k == L) | (k == TT)).any() or (int((k == 0).sum()) >= rq and not exhausted):
                    break
            P["C"] += 1; Ln["C"] += int(node.sum()); pos[node] += 1
            k = _kind(ev, ln, stride, ray, pos)
    return P, Ln


def replay_rays_per_lane(ev, ln, node_q, leaf_q, idle_q, R, sw_min):
    n = len(ln); stride = ev.shape[1]; ln = np.minimum(ln, stride)
    ray = np.full((R, 64), -1, np.int64); pos = np.zeros((R, 64), np.int64); cur = np.zeros(64, np.int64); nxt = 0
    P = dict(A=0, B=0, C=0, S=0); Ln = dict(A=0, B=0, C=0, S=0)
    ar = np.arange(64)

    def serve(want):
        nonlocal cur
        cw = want[cur, ar]
        need_sw = want.any(axis=0) & ~cw
        if need_sw.sum() >= sw_min or not cw.any():
            cur = np.where(need_sw, want.argmax(axis=0), cur)
            P["S"] += 1; Ln["S"] += int(need_sw.sum())
            cw = want[cur, ar]
        return cw

    while True:
        k = _kind(ev, ln, stride, ray, pos)
        idle = k == 0
        n_idle = int(idle.sum()) if nxt < n else 0
        isleaf = (k == L) | (k == TT); isnode = k == N
        nB = int(isleaf.any(axis=0).sum()); nC = int(isnode.any(axis=0).sum())
        if n_idle == 0 and nB == 0 and nC == 0:
            break
        if n_idle and (n_idle >= idle_q or nC < node_q):
            for s in range(R):
                need = np.where(idle[s])[0]
                if not len(need) or nxt >= n:
                    continue
                take = min(len(need), n - nxt)
                P["A"] += 1; Ln["A"] += len(need)
                ray[s, need] = -1; ray[s, need[:take]] = np.arange(nxt, nxt + take); pos[s, need[:take]] = 0; nxt += take
            continue
        if nB and (nB >= leaf_q or nC < node_q):
            while True:
                m = serve(isleaf)
                P["B"] += 1; Ln["B"] += int(m.sum()); pos[cur[m], ar[m]] += 1
                k = _kind(ev, ln, stride, ray, pos); isleaf = (k == L) | (k == TT)
                if int(isleaf.any(axis=0).sum()) < leaf_q:
                    break
            continue
        while True:
            isnode = k == N; nC = int(isnode.any(axis=0).sum())
            if nC == 0:
                break
            if nC < node_q and (((k == L) | (k == TT)).any() or ((k == 0).any() and nxt < n)):
                break
            m = serve(isnode)
            P["C"] += 1; Ln["C"] += int(m.sum()); pos[cur[m], ar[m]] += 1
            k = _kind(ev, ln, stride, ray, pos)
    return P, Ln


def report(name, fn, queues, sw=0):
    for flavour in ("closest", "shadow"):
        rays = sum(len(ln) for ev, ln in queues[flavour])
        P = dict(A=0, B=0, C=0, S=0); Ln = dict(A=0, B=0, C=0, S=0)
        for ev, ln in queues[flavour]:
            p, l = fn(ev, ln)
            for key in p:
                P[key] += p[key]; Ln[key] += l[key]
        instr = (sum(COST[k] * P[k] for k in "ABC") + sw * P["S"]) / rays
        print("%-44s %-8s passes per ray (lanes per pass): A %.3f (%4.1f)  B %.3f (%4.1f)  C %.3f (%4.1f)  switch %.3f (%4.1f) -> %6.1f vector instructions per ray"
              % (name, flavour, P["A"] / rays, Ln["A"] / max(P["A"], 1), P["B"] / rays, Ln["B"] / max(P["B"], 1), P["C"] / rays, Ln["C"] / max(P["C"], 1),
                 P["S"] / rays, Ln["S"] / max(P["S"], 1), instr), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays-per-queue", type=int, default=4000)
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--height", type=int, default=90)
    a = ap.parse_args()
    sys.argv = ["wave_schedule_model", "--rays-per-queue", str(a.rays_per_queue), "--width", str(a.width), "--height", str(a.height)]
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        q = M.main()
    def shipped(ev, ln):
        p, l = M.replay(ev, ln, 32, 8)
        p["S"] = 0; l["S"] = 0
        return p, l
    report("shipped (node_q 32 : leaf_q 8)", shipped, q)
    report("greedy by lanes served per instruction", replay_greedy, q)
    for nq, lq, rq in ((32, 8, 8), (32, 8, 16), (32, 8, 24), (40, 12, 16)):
        report("refill quorum %d at %d:%d" % (rq, nq, lq), lambda ev, ln: replay_quorum(ev, ln, nq, lq, rq), q)
    for sw in (40, 25):
        for nq, lq, iq, swm in ((48, 24, 16, 1), (40, 16, 8, 1), (48, 24, 16, 8), (48, 24, 16, 16), (56, 32, 24, 12)):
            report("2 rays / lane, %d:%d, idle %d, swmin %d, SW %d" % (nq, lq, iq, swm, sw), lambda ev, ln: replay_rays_per_lane(ev, ln, nq, lq, iq, 2, swm), q, sw)
    report("3 rays / lane, 56:32, idle 24, swmin 8, SW 40", lambda ev, ln: replay_rays_per_lane(ev, ln, 56, 32, 24, 3, 8), q, 40)


if __name__ == "__main__":
    main()
