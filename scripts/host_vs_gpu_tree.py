import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
rt = ge.load_package(); rt.init(0)
cfg = rt.ParsedConfigState().load_config_file("tests/golden/final_avec_bonus.scene")
cfg.set_size(3840, 2160); f = cfg.frame()
for gb in (False, True):
    info = rt.upload_scene(cfg.scene(), gpu_build=gb)
    ts = [rt.render_frame(f, want_stats=True)[1]["kernel_ms"] for _ in range(6)]
    print("gpu_build", gb, "build_ms %.1f" % info["build_ms"], "sah %.2f" % info["sah_cost"], " ".join("%.3f" % t for t in ts))
