"""Counters of the instrumented kernel for the 10 M-triangle mesh at 7680x4320, whole and by maxdepth (diagnostic)."""
import os, sys, importlib, json
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import __graft_entry__ as ge
rt = ge.load_package(); rt.init(0)
syn = importlib.import_module("raytracer_rust_amd.synthetic")
scene = syn.product_scene(rt); rt.upload_scene(scene.c, gpu_build=True); f = syn.product_frame(rt, 7680, 4320)
st = rt.render_frame(f, want_stats=True, count_work=True)[1]
print(json.dumps(st))
for md in (1,2,3):
    f.maxdepth = md
    s = rt.render_frame(f, count_work=True)[1]
    ts=[rt.render_frame(f, want_stats=True)[1]["kernel_ms"] for _ in range(4)]
    print(md, min(ts), s["nodes_visited"], s["prims_tested"], s["box_passes"], s["shape_passes"], s["shadow"], s["secondary"], s["phase_passes"])
