#!/usr/bin/env python3
"""Renders a benchmark workload a few times on cuda:0 through the C ABI.  Small driver for rocprofv3 passes:
  rocprofv3 --pmc ... -- python3 scripts/render_once.py [frames] [W H] [maxdepth]      the dragon (default 3840x2160)
  rocprofv3 --pmc ... -- python3 scripts/render_once.py [frames] mesh                  the 10 M-triangle mesh at 7680x4320"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rt = ge.load_package()
rt.init(0)
if len(sys.argv) > 2 and sys.argv[2] == "mesh":
    syn = importlib.import_module("raytracer_rust_amd.synthetic")
    scene = syn.product_scene(rt)
    rt.upload_scene(scene.c, gpu_build=True)
    f = syn.product_frame(rt, 7680, 4320)
else:
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(ROOT, "tests", "golden", "final_avec_bonus.scene"))
    cfg.set_size(w, h)
    rt.upload_scene(cfg.scene())
    f = cfg.frame()
    if len(sys.argv) > 4:
        f.maxdepth = int(sys.argv[4])
for _ in range(frames):
    _, st = rt.render_frame(f, want_stats=True)
print("kernel_ms", st["kernel_ms"])
