import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import __graft_entry__ as ge, numpy as np
rt = ge.load_package(); rt.init(0)
cfg = rt.ParsedConfigState().load_config_file(os.path.join(os.path.dirname(ge.__file__), "tests", "golden", "final_avec_bonus.scene"))
cfg.set_size(3840, 2160); f = cfg.frame()
for rep in range(3):
    rt.upload_scene(cfg.scene())
    ts = [rt.render_frame(f, want_stats=True)[1] for _ in range(4)]
    print("cold %.3f total %.3f | then" % (ts[0]["kernel_ms"], ts[0]["total_ms"]), " ".join("%.3f" % t["kernel_ms"] for t in ts[1:]), flush=True)
