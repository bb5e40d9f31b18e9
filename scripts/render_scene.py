#!/usr/bin/env python3
"""What the reference's main does (src/main.rs:7-27: load the scene file, build the tracer, render, save the PNG
next to the name the scene file gives), on one MI355X, with the time of every stage.

  python scripts/render_scene.py path/to/file.scene [-o out.png] [--size W H] [--gpu-build] [--frames N]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("-o", "--output", default=None, help="PNG to write (default: the scene file's `output` name in the current directory)")
    ap.add_argument("--size", type=int, nargs=2, metavar=("W", "H"), default=None, help="override the scene file's size line")
    ap.add_argument("--gpu-build", action="store_true", help="build the tree on the GPU")
    ap.add_argument("--frames", type=int, default=1, help="render the frame this many times (the last one is saved)")
    args = ap.parse_args()
    rt = ge.load_package()
    t = [time.perf_counter()]

    def lap():
        t.append(time.perf_counter())
        return 1e3 * (t[-1] - t[-2])

    rt.init(0)
    ms_init = lap()
    cfg = rt.ParsedConfigState().load_config_file(args.scene)
    if args.size:
        cfg.set_size(*args.size)
    ms_parse = lap()
    tracer = rt.RayTracer(cfg, gpu_build=args.gpu_build)
    ms_build = lap()
    kernel_ms = []
    for _ in range(max(1, args.frames)):
        img = tracer.render()
        kernel_ms.append(tracer.last_stats["kernel_ms"])
    ms_render = lap()
    out = args.output or os.path.basename(tracer.get_output_path())
    rt.save_image(img, out)
    ms_save = lap()
    info = tracer.bvh_info
    print(f"{args.scene}: {img.width}x{img.height}, {info['bvh_prims']} shapes in the tree ({info['node_pairs']} node pairs, depth {info['max_depth']}), "
          f"{info['list_prims']} outside it")
    print(f"  GPU start-up {ms_init:9.1f} ms\n  parse        {ms_parse:9.1f} ms\n  tree + upload{ms_build:9.1f} ms ({'GPU' if args.gpu_build else 'host'} builder)\n"
          f"  render       {ms_render:9.1f} ms ({len(kernel_ms)} frame(s); kernel {min(kernel_ms):.3f} ms best, {kernel_ms[0]:.3f} ms first; the rest is the copy to the host)\n"
          f"  PNG          {ms_save:9.1f} ms -> {out}")


if __name__ == "__main__":
    main()
