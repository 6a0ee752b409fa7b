"""Would path regeneration pay?  A model of VERDICT r01 item 5 on the headline workload, from measured quantities only.

Two ways to bound the per-path device memory of rt_integrate to G GiB for a 1920x1080, 8-bounce frame:

  chunked       what RT_OPT_PATH_STATE_LIMIT_MB does today: S = G / (540 B x pixels) samples of every pixel travel
                through the wavefront loop together; launch b of a batch holds the survivors of bounce b, so the nine
                closest-hit launches shrink from P to 0.064 P rays and each pays the constant tail.
  regeneration  a pool of R ray slots (240 B: ping-pong ray queues, hits, double-buffered shadow queue) that is refilled
                every round with the next samples of pixels that have room, so every launch is full; the radiance log
                (292 B per sample: 4 + 18 x 16) must keep a sample's contributions until ALL earlier samples of its pixel
                have retired -- the fp32 sum has to be associated in sample order to stay bit-identical to the reference
                 -- so a window of W log slots per pixel works like a reorder buffer.

Inputs: per-bounce survival of the benchmark scene (instrumented oracle, 480x270, 2 spp: /tmp-free, printed by
tests/_oracle.Oracle.last_counts), the launch-time law measured on MI355X (profiles/r02_samples_in_flight_sweep.log:
a closest-hit launch of R rays takes 0.77 ms + R / 6.8 Grays/s; shadow launches 0.77 ms + R / 10.4 G, about half of
their constant hidden by RT_OPT_OVERLAP_SHADOW), k_shade at 0.057 ns per queue entry.  The reorder buffer is simulated
(in-order retire per pixel, one bounce per round, global ray-slot cap).

usage: python tools/regeneration_model.py [GiB ...]"""
import sys
import numpy as np

PIXELS = 1920 * 1080
CLOSEST = np.array([1.0, 0.9744, 0.5088, 0.3052, 0.2017, 0.1447, 0.1075, 0.0823, 0.0640])   # rays traced at bounce b per path
SHADOW = np.array([0.8522, 0.3502, 0.2054, 0.1262, 0.0879, 0.0625, 0.0470, 0.0347, 0.0278])
TAIL_MS, CLOSEST_GRAYS, SHADOW_GRAYS, SHADE_NS = 0.77, 6.8, 10.4, 0.057
SHADOW_TAIL_EXPOSED = 0.5
PATH_BYTES, LOG_BYTES, RAY_BYTES = 540.0, 292.0, 240.0


def ms_closest(rays):
    return TAIL_MS + rays / (CLOSEST_GRAYS * 1e6)


def ms_shadow(rays):
    return SHADOW_TAIL_EXPOSED * TAIL_MS + rays / (SHADOW_GRAYS * 1e6)


def chunked(gib):
    """ms per sample per pixel of the whole frame with S samples in flight (the frame in one chunk)."""
    s = max(int(gib * 2 ** 30 / (PATH_BYTES * PIXELS)), 1)
    p = s * PIXELS
    ms = sum(ms_closest(p * f) for f in CLOSEST) + sum(ms_shadow(p * g) for g in SHADOW) + p * CLOSEST.sum() * SHADE_NS * 1e-6
    return s, ms / s


def regeneration(gib, window, rng, pixels=4096, rounds=600):
    """Reorder-buffer simulation: `window` log slots per pixel, the rest of the memory as ray slots."""
    ray_slots_per_pixel = (gib * 2 ** 30 / PIXELS - window * LOG_BYTES) / RAY_BYTES
    if ray_slots_per_pixel < 0.5:
        return None
    cap = int(ray_slots_per_pixel * pixels)
    # path length L (closest-hit rays traced): P(L > b) = CLOSEST[b]
    surv = np.append(CLOSEST, 0.0)
    pmf = surv[:-1] - surv[1:]
    # state per pixel: ring of `window` samples, remaining bounces (0 = finished, waiting to retire; -1 = free)
    rem = -np.ones((pixels, window), np.int32)
    head = np.zeros(pixels, np.int64)           # next sample to retire
    issued = np.zeros(pixels, np.int64)         # next sample to generate
    launches, rays_total, retired = 0, 0, 0
    warm = rounds // 3
    for r in range(rounds):
        # retire in order
        while True:
            slot = head % window
            done = (rem[np.arange(pixels), slot] == 0) & (head < issued)
            if not done.any():
                break
            rem[np.arange(pixels)[done], slot[done]] = -1
            head[done] += 1
            if r >= warm:
                retired += int(done.sum())
        # generate into free log slots while ray slots last (pixel order rotates so that no pixel starves)
        live = int((rem > 0).sum())
        room = cap - live
        order = np.roll(np.arange(pixels), -(r * 977) % pixels)
        for _ in range(window):
            if room <= 0:
                break
            can = (issued - head < window)
            idx = order[can[order]][:room]
            if len(idx) == 0:
                break
            rem[idx, issued[idx] % window] = rng.choice(np.arange(1, 10), size=len(idx), p=pmf)
            issued[idx] += 1
            room -= len(idx)
        # one round = one closest-hit launch over every live ray (+ its shade and shadow launches)
        live = int((rem > 0).sum())
        if r >= warm:
            launches += 1
            rays_total += live
        rem[rem > 0] -= 1
    scale = PIXELS / pixels
    rays_per_launch = rays_total / launches * scale
    shadow_per_closest = SHADOW.sum() / CLOSEST.sum()
    ms_round = ms_closest(rays_per_launch) + ms_shadow(rays_per_launch * shadow_per_closest) + rays_per_launch * SHADE_NS * 1e-6
    samples_per_round = retired / launches * scale / PIXELS
    return dict(window=window, ray_slots_per_pixel=round(ray_slots_per_pixel, 1), fill=round(rays_total / launches / cap, 3),
                Mrays_per_launch=round(rays_per_launch / 1e6, 1), ms_per_spp=ms_round / samples_per_round)


def main():
    rng = np.random.default_rng(1)
    rays_per_spp = PIXELS * (CLOSEST.sum() + SHADOW.sum())
    print("rays per sample per pixel of the frame: %.2f M (%.2f closest + %.2f shadow per path)" % (rays_per_spp / 1e6, CLOSEST.sum(), SHADOW.sum()))
    for gib in [float(a) for a in sys.argv[1:]] or [8.0, 16.0, 32.0, 64.0, 133.0]:
        s, ms = chunked(gib)
        line = "%6.1f GiB | chunked: %3d samples in flight, %.3f ms/spp = %4.0f Mrays/s" % (gib, s, ms, rays_per_spp / ms / 1e3)
        best = None
        for w in (4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192):
            if w * LOG_BYTES * PIXELS > gib * 2 ** 30:
                break
            res = regeneration(gib, w, rng)
            if res and (best is None or res["ms_per_spp"] < best["ms_per_spp"]):
                best = res
        if best:
            line += " | regeneration (best window %d, %.1f ray slots per pixel, launches %.0f %% full, %.1f M rays each): %.3f ms/spp = %4.0f Mrays/s (%+.1f %%)" % (
                best["window"], best["ray_slots_per_pixel"], 100 * best["fill"], best["Mrays_per_launch"], best["ms_per_spp"],
                rays_per_spp / best["ms_per_spp"] / 1e3, 100 * (ms / best["ms_per_spp"] - 1))
        print(line)


if __name__ == "__main__":
    main()
