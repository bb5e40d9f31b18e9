#!/usr/bin/env python3
"""Reads the per-tile wavefront durations / start times left in the frame by a library built with -DRT_WAVE_TIMES
(diagnostic, not part of the product): occupancy over time and where the slow tiles are."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
rt = ge.load_package()
cfg = rt.ParsedConfigState().load_config_file(os.path.join(ROOT, "tests", "golden", "final_avec_bonus.scene"))
cfg.set_size(3840, 2160)
rt.RayTracer(cfg)
f = cfg.frame()
rt.render(f); rt.render(f); rt.render(f)
img, st = rt.render(f)
np.save(os.path.join(ROOT, "gpurun_out", f"wave_{sys.argv[1]}.npy"), img[::8, ::8].copy())
print(sys.argv[1], "kernel ms", st["kernel_ms"], "min/max", int(img.min()), int(img.max()))
