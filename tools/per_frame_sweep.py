"""Sweep of the library's small-launch settings on the reference's own call pattern: one Integrate() per frame, one
sample per pixel in flight (bench.py's `per_frame` leg), scene built once.
usage: python tools/per_frame_sweep.py [--config 4] [--frames 32] [--settings name:tune:small_launch_paths:overlap:variant[:resolve] ...]
  tune = RT_OPT_TRACE_TUNE (hex ok; bits 24..31 = fewest rays per lane a wave of k_trace_w4's grid is started for),
  small_launch_paths = RT_OPT_SMALL_LAUNCH_PATHS, overlap = RT_OPT_OVERLAP_SHADOW, variant = RT_OPT_TRACE_VARIANT"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--settings", nargs="*", default=["default:0:2000000:1:5"])
    a = ap.parse_args()
    from raytracing_amd import capi, host, scenes as S
    cfg = bench.CONFIGS[a.config]
    args = argparse.Namespace(config=a.config, scene=None, blob_tris=871_200, ball_tris=20_000, width=a.width or cfg["width"],
                              height=a.height or cfg["height"], bounces=cfg["bounces"])
    scene, n_tris = bench.build_scene(args, host, S)
    render = host.Render(args.width, args.height, scene)
    render.set_camera(host.default_camera(args.width, args.height))
    render.set_max_bounces(args.bounces)
    frame = host.load().rth_render_frame_handle(render.handle)
    lib = capi.load()
    for spec in a.settings:
        parts = spec.split(":")
        name, tune, small, overlap, variant = parts[:5]
        resolve = int(parts[5]) if len(parts) > 5 else 1            # 0: no ResolveRadiance + read-back per frame (diagnostic)
        assert lib.rt_set_option(frame, capi.OPT_TRACE_TUNE, int(tune, 0)) == 0
        assert lib.rt_set_option(frame, capi.OPT_SMALL_LAUNCH_PATHS, int(small)) == 0
        assert lib.rt_set_option(frame, capi.OPT_OVERLAP_SHADOW, int(overlap)) == 0
        assert lib.rt_set_option(frame, capi.OPT_TRACE_VARIANT, int(variant)) == 0
        pf = bench.per_frame_leg(args, render, lib, frame, capi, a.frames, resolve=bool(resolve))
        print("%-28s %8.1f Mrays/s  %8.3f ms/frame  (%d x %d, %d bounces, %d tris)" %
              (name, pf["mrays_per_s"], pf["ms_per_frame"], args.width, args.height, args.bounces, n_tris), flush=True)


if __name__ == "__main__":
    main()
