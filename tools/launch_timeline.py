"""Where a closest-hit launch spends its time: k_trace_w4 records when its first wave started, when the
first wave found the ray queue dry and when its last wave left (rt_frame_debug_timeline, 100 MHz wall
clock).  ramp = start -> dry is the phase with every wave fed; drain = dry -> end is the tail in which
waves finish the rays they hold and the machine empties.  One batch of S samples per pixel per row."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import capi, host, scenes as S

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--in-flight", default="4,16,64,128")
ap.add_argument("--overlap-shadow", type=int, default=0, help="1: the shadow trace runs beside the launches measured")
ap.add_argument("--tune", type=lambda x: int(x, 0), default=0)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
scene, n_tris = bench.build_scene(argparse.Namespace(config=a.config, blob_tris=871_200, ball_tris=20_000), host, S)
lib = capi.load()
for s in [int(x) for x in a.in_flight.split(",")]:
    render = host.Render(cfg["width"], cfg["height"], scene)
    render.set_camera(host.default_camera(cfg["width"], cfg["height"]))
    render.set_max_bounces(cfg["bounces"])
    render.set_resolve_every_frame(False)
    frame = host.load().rth_render_frame_handle(render.handle)
    assert lib.rt_set_option(frame, capi.OPT_SAMPLES_IN_FLIGHT, s) == 0
    assert lib.rt_set_option(frame, capi.OPT_OVERLAP_SHADOW, a.overlap_shadow) == 0
    if a.tune:
        assert lib.rt_set_option(frame, capi.OPT_TRACE_TUNE, a.tune) == 0
    render.reserve_samples(s)
    render.render_samples(s); render.finish()
    assert lib.rt_frame_debug_timeline(frame, 1, None) == 0
    st0 = render.stats()
    render.render_samples(s); render.finish()
    st = render.stats()
    out = (C.c_ulonglong * 448)()
    assert lib.rt_frame_debug_timeline(frame, 0, out) == 0
    print("samples in flight %d (%d closest rays in the batch)" % (s, st.closest_rays - st0.closest_rays))
    tot_run = tot_drain = 0.0
    for b in range(64):
        t0, td, t1, max_steps, slow_ticks, slow_steps = [out[6 * b + k] for k in range(6)]
        if t1 == 0:
            continue
        run, drain = (t1 - t0) / 100.0, ((t1 - td) / 100.0 if td else 0.0)
        tot_run += run; tot_drain += drain
        print("  bounce %d: launch %8.1f us, fed %8.1f us, drain %7.1f us (%.0f %%); slowest ray %7.1f us for %d steps (%.2f us/step), most steps %d"
              % (b, run, run - drain, drain, 100 * drain / run, slow_ticks / 100.0, slow_steps, slow_ticks / 100.0 / max(slow_steps, 1), max_steps))
    print("  all bounces: %.2f ms in closest launches, %.2f ms of it draining (%.0f %%)" % (tot_run / 1e3, tot_drain / 1e3, 100 * tot_drain / tot_run))
    hist = [out[384 + i] for i in range(64)]
    total = float(sum(hist)) or 1.0
    acc, line = 0, []
    for i, n in enumerate(hist):
        acc += n
        if i % 4 == 3 or i == 63:
            line.append("%d us %.1f %%" % ((i + 1) * 25, 100.0 * acc / total))
    print("  waves gone by (after the queue ran dry): " + ", ".join(line[:12]) + ", ..., " + line[-1])
    del render
