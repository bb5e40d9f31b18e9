#!/usr/bin/env python3
"""Host time of one rt_render_frame call (enqueue only, nothing waited for), plain and through the in-library multi-GPU path
with a communicator of one: what one host thread spends per frame and GPU, which bounds the frame rate of rt_init_multi(n)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
rt = ge.load_package()
cfg = rt.ParsedConfigState().load_config_file(os.path.join(ROOT, "tests", "golden", "final_avec_bonus.scene"))
cfg.set_size(3840, 2160); f = cfg.frame()
for mode in ("single", "multi1"):
    rt.init(0) if mode == "single" else rt.init_multi(1)
    rt.upload_scene(cfg.scene())
    for _ in range(5): rt.render_frame(f)
    rt.synchronize()
    for n in (4, 16):
        t0 = time.perf_counter()
        for _ in range(n): rt.render_frame(f)
        t1 = time.perf_counter()
        rt.synchronize()
        t2 = time.perf_counter()
        print(f"{mode}: {n} frames enqueued in {(t1 - t0) * 1e6 / n:.1f} us each (python call included), done after {(t2 - t0) * 1e3 / n:.3f} ms each", flush=True)
    rt.shutdown()
