"""How full are k_trace_w4's waves?  A lane-by-lane replay of the kernel's wave-level schedule (trace_kernels.h: phase A
retire / refill, loop B leaves with its `leaf_q` threshold, loop C wide nodes with its `node_q` threshold) over the step
sequences of real rays -- the walk restated on the CPU (oracle/oracle.c: orc_wide_trace_events) on the benchmark scene's
closest-hit and shadow queues.  Counts passes and active lanes per pass, for the shipped thresholds, for others, and for the
bound no lane-bound schedule can beat (every pass full: rays, not lanes, resident -- DESIGN.md section 6b).  No GPU.
usage: python tools/wave_schedule_model.py [--triangles 2800000] [--width 320 --height 180]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracing_amd import host, scenes as S, types as T
from tests import _oracle
from tests.test_wide_bvh import wide_of

N, L, TT = ord("N"), ord("L"), ord("T")


def replay(ev, ln, node_q, leaf_q, direct_steps=None, refill_q=1):
    """One persistent wave draining the queue `ev` (rays x steps).  Returns passes and lane-steps per loop."""
    n = len(ln)
    stride = ev.shape[1]
    ln = np.minimum(ln, stride)
    ray = np.full(64, -1, np.int64)            # ray held by each lane
    pos = np.zeros(64, np.int64)               # next step of that ray
    nxt = 0
    passes = dict(A=0, B=0, C=0)
    lanes = dict(A=0, B=0, C=0)
    def kind():
        k = np.zeros(64, np.int64)             # 0 idle, N, L/T
        have = ray >= 0
        idx = np.where(have)[0]
        if len(idx):
            k[idx] = ev[ray[idx], np.minimum(pos[idx], stride - 1)]
            k[idx[pos[idx] >= ln[ray[idx]]]] = 0
        return k
    while True:
        k = kind()
        idle = k == 0
        # (refill_q: phase A waits until that many lanes are idle, unless the node loop has nothing to run on -- a policy question
        # this model can price: --refill-quorum)
        if idle.any() and (int(idle.sum()) >= refill_q or int((k == N).sum()) < node_q or nxt >= n):
            # A: retire finished rays, hand out new ones
            need = np.where(idle)[0]
            take = min(len(need), n - nxt)
            if take or (ray[need] >= 0).any():
                passes["A"] += 1
                lanes["A"] += int(len(need))
            ray[need] = -1
            if take:
                ray[need[:take]] = np.arange(nxt, nxt + take)
                pos[need[:take]] = 0
                nxt += take
            k = kind()
        exhausted = nxt >= n
        if not (k != 0).any():
            if exhausted:
                break
            continue
        # B: leaves
        leaf = (k == L) | (k == TT)
        n_node = int((k == N).sum())
        if leaf.any() and (int(leaf.sum()) >= leaf_q or n_node < node_q):
            while True:
                passes["B"] += 1
                lanes["B"] += int(leaf.sum())
                pos[leaf] += 1
                k = kind()
                leaf = (k == L) | (k == TT)
                if not (leaf.any() and int(leaf.sum()) >= leaf_q):
                    break
        # C: wide nodes
        while True:
            node = k == N
            if not node.any():
                break
            if int(node.sum()) < node_q:
                waiting = ((k == L) | (k == TT)).any() or ((k == 0).any() and not exhausted)
                if waiting:
                    break
            passes["C"] += 1
            lanes["C"] += int(node.sum())
            pos[node] += 1
            k = kind()
    return passes, lanes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--triangles", type=int, default=2_800_000)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=180)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--rays-per-queue", type=int, default=12000, help="rays of each queue the model wave drains")
    ap.add_argument("--refill-quorum", default="1", help="comma-separated quorums of phase A to price at 32:8 (1 = the kernel's rule: any idle lane)")
    a = ap.parse_args()
    scene = host.Scene(arrays=S.city_block(a.triangles))
    scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    scene.set_env_path(os.path.join(ROOT, "assets", "ibl", "CGSkies_0036_free.hdr"))
    scene.build_bvh(); scene.finalize()
    arrays = scene.arrays()
    wide, entry = wide_of(arrays["nodes"])
    w, h, n = a.width, a.height, a.width * a.height
    orc = _oracle.Oracle(w, h, arrays)
    orc.set_camera(host.default_camera(w, h)); orc.set_max_bounces(a.bounces)
    orc.stage("reset"); orc.stage("generate_rays")
    queues = {"closest": [], "shadow": []}
    for bounce in range(a.bounces + 1):
        k = int(orc.buffer("ray_counter%d" % (bounce & 1), np.uint32, 1)[0])
        rays = orc.buffer("rays%d" % (bounce & 1), T.ray, n)[:k]
        queues["closest"].append(orc.wide_trace_events(wide, entry, rays[: a.rays_per_queue], False))
        orc.stage("intersect", bounce)
        for st, args in (("shade_miss", (bounce,)), ("clear_counters", (bounce,)), ("shade_hits", (bounce,))):
            orc.stage(st, *args)
        ks = int(orc.buffer("shadow_ray_counter", np.uint32, 1)[0])
        srays = orc.buffer("shadow_rays", T.ray, n)[:ks]
        queues["shadow"].append(orc.wide_trace_events(wide, entry, srays[: a.rays_per_queue], True))
        orc.stage("intersect_shadow"); orc.stage("accumulate")
    for flavour in ("closest", "shadow"):
        steps = sum(int(np.minimum(ln, ev.shape[1]).sum()) for ev, ln in queues[flavour])
        rays = sum(len(ln) for ev, ln in queues[flavour])
        nodes = sum(int((ev[:, :] == N).sum()) for ev, ln in queues[flavour])
        print("%s: %d rays, %.1f steps per ray (%.1f wide nodes, %.1f leaf / triangle passes); a schedule with every pass full needs %.3f passes per ray"
              % (flavour, rays, steps / rays, nodes / rays, (steps - nodes) / rays, steps / rays / 64.0))
        for node_q, leaf_q in ((32, 8), (24, 8), (40, 8), (32, 16), (32, 4), (48, 16), (16, 4), (1, 1), (64, 64)):
            P = dict(A=0, B=0, C=0); Ln = dict(A=0, B=0, C=0)
            for ev, ln in queues[flavour]:
                p, l = replay(ev, ln, node_q, leaf_q)
                for key in P:
                    P[key] += p[key]; Ln[key] += l[key]
            tot = P["B"] + P["C"]
            print("   node_q:leaf_q %2d:%-2d  passes per ray: C %.3f (%.1f lanes) + B %.3f (%.1f lanes) + A %.3f = %.3f B+C -> lane utilisation %.0f %%, %.2f x the full-pass bound"
                  % (node_q, leaf_q, P["C"] / rays, Ln["C"] / max(P["C"], 1), P["B"] / rays, Ln["B"] / max(P["B"], 1), P["A"] / rays, tot / rays,
                     100.0 * (Ln["B"] + Ln["C"]) / max(tot, 1) / 64.0, tot / (steps / 64.0)))


    return queues


def price_refill(queues, quorums, cost=dict(A=220, B=140, C=179)):
    """vector instructions per ray (static per-pass counts of the three loops, tools/isa_mix.py --loops) for each refill quorum"""
    for flavour in ("closest", "shadow"):
        rays = sum(len(ln) for ev, ln in queues[flavour])
        for rq in quorums:
            P = dict(A=0, B=0, C=0)
            for ev, ln in queues[flavour]:
                p, l = replay(ev, ln, 32, 8, refill_q=rq)
                for key in P:
                    P[key] += p[key]
            print("%s  refill quorum %2d: passes per ray A %.3f B %.3f C %.3f -> %.1f vector instructions per ray" % (
                flavour, rq, P["A"] / rays, P["B"] / rays, P["C"] / rays, sum(cost[k] * P[k] for k in P) / rays))


if __name__ == "__main__":
    q = main()
    rq = [int(x) for x in sys.argv[sys.argv.index("--refill-quorum") + 1].split(",")] if "--refill-quorum" in sys.argv else []
    if rq:
        price_refill(q, rq)
