# Frames per second of the 4K dragon with 1, 2 and 4 frames in flight on one GPU (the library's lanes).
import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import __graft_entry__ as ge
rt = ge.load_package(); rt.init(0)
cfg = rt.ParsedConfigState().load_config_file(os.path.join(os.path.dirname(ge.__file__), "tests", "golden", "final_avec_bonus.scene"))
cfg.set_size(3840, 2160); f = cfg.frame()
rt.upload_scene(cfg.scene())
for lanes in (1, 2, 4):
    for warm in range(8):
        rt.select_lane(warm % lanes); rt.render_frame(f)
    for l in range(lanes):
        rt.select_lane(l); rt.synchronize()
    n = 400
    t0 = time.perf_counter()
    for i in range(n):
        rt.select_lane(i % lanes); rt.render_frame(f)
    for l in range(lanes):
        rt.select_lane(l); rt.synchronize()
    dt = time.perf_counter() - t0
    print(f"{lanes} in flight: {dt / n * 1e3:.3f} ms per frame, {n / dt:.1f} frames/s", flush=True)
rt.select_lane(0)
