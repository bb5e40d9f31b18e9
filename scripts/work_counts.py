# Counters of the instrumented kernel for the 4K dragon frame: boxes and shapes tested, rays by kind, wavefront passes.
import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import __graft_entry__ as ge
rt = ge.load_package(); rt.init(0)
cfg = rt.ParsedConfigState().load_config_file(os.path.join(os.path.dirname(ge.__file__), "tests", "golden", "final_avec_bonus.scene"))
cfg.set_size(3840, 2160); f = cfg.frame()
rt.upload_scene(cfg.scene())
st = rt.render_frame(f, want_stats=True, count_work=True)[1]
print(st)
import json
os.makedirs("gpurun_out", exist_ok=True)
json.dump(st, open(os.path.join("gpurun_out", "work_counts.json"), "w"))      # input of scripts/phase_table.py --counts
