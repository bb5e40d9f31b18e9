"""Writes the measured costs of the tile regions of the 4K dragon frame (cold launch and an ordered one) to gpurun_out/."""
import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import __graft_entry__ as ge, numpy as np
rt = ge.load_package(); rt.init(0)
root = os.path.dirname(ge.__file__)
cfg = rt.ParsedConfigState().load_config_file(os.path.join(root, "tests", "golden", "final_avec_bonus.scene"))
cfg.set_size(3840, 2160); f = cfg.frame()
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
rt.upload_scene(cfg.scene())
out = {}
for k in range(4):
    _, st = rt.render_frame(f, want_stats=True)
    out["costs%d" % k] = rt.tile_costs_export(f)
    print("frame %d kernel %.3f ms" % (k, st["kernel_ms"]), flush=True)
np.savez(os.path.join(root, "gpurun_out", "dragon_tile_costs.npz"), **out)
