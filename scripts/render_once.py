#!/usr/bin/env python3
"""Renders the headline workload (dragon at 3840x2160) a few times on cuda:0 through the C ABI.
Small driver for rocprofv3 passes: `rocprofv3 --pmc ... -- python3 scripts/render_once.py [frames] [W H] [maxdepth]`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 5
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
rt = ge.load_package()
cfg = rt.ParsedConfigState().load_config_file(os.path.join(ROOT, "tests", "golden", "final_avec_bonus.scene"))
cfg.set_size(w, h)
tracer = rt.RayTracer(cfg, device=0)
f = cfg.frame()
if len(sys.argv) > 4:
    f.maxdepth = int(sys.argv[4])
for _ in range(frames):
    _, st = rt.render(f)
print("kernel_ms", st["kernel_ms"])
