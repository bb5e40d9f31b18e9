#!/usr/bin/env python3
"""Kernel time and ray counts of the dragon frame as a function of maxdepth (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
rt = ge.load_package()
cfg = rt.ParsedConfigState().load_config_file(os.path.join(ROOT, "tests", "golden", "final_avec_bonus.scene"))
cfg.set_size(3840, 2160)
tracer = rt.RayTracer(cfg, device=0)
f = cfg.frame()
for md in (1, 2, 3, 4, 5):
    f.maxdepth = md
    _, st = rt.render_frame(f, count_work=True)      # (frames left on the device: rt.render's stores into host memory would bound the short ones)
    ts = []
    for _ in range(8):
        _, s = rt.render_frame(f, want_stats=True)
        ts.append(s["kernel_ms"])
    rays = st["primary"] + st["shadow"] + st["secondary"]
    print(f"maxdepth {md}: kernel {min(ts):.3f} ms  rays {rays/1e6:.2f}M (p {st['primary']/1e6:.2f} s {st['shadow']/1e6:.2f} r {st['secondary']/1e6:.2f}) "
          f"boxes {st['nodes_visited']/1e6:.1f}M shapes {st['prims_tested']/1e6:.1f}M  {rays/min(ts)/1e3:.0f} Mrays/s  "
          f"lane use: box loop {st['nodes_visited'] / 2 / (64 * max(st['box_passes'], 1)):.3f} ({st['box_passes']/1e6:.1f}M passes), "
          f"shape loop {st['prims_tested'] / (64 * max(st['shape_passes'], 1)):.3f} ({st['shape_passes']/1e6:.1f}M passes)")
