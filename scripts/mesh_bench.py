#!/usr/bin/env python3
"""Kernel time and counters of the large procedural mesh at 7680x4320 and of the 4K dragon (diagnostic)."""
import os, sys, time, zlib, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
rt = ge.load_package()
rt.init(0)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RT_"))

def run(name, frame):
    img, st = rt.render(frame, count_work=True)
    ts = []
    for _ in range(6):
        _, s = rt.render(frame)
        ts.append(s["kernel_ms"])
    rays = st["primary"] + st["shadow"] + st["secondary"]
    gb = (32 * st["nodes_visited"] + 48 * st["prims_tested"] + 4 * st["primary"]) / 1e9
    print(f"[{tag}] {name}: kernel {min(ts):.3f} ms  crc {zlib.crc32(img.tobytes()):08x}  rays {rays/1e6:.1f}M  {rays/min(ts)/1e3:.0f} Mrays/s  "
          f"alg {gb:.1f} GB -> {gb/min(ts):.2f} TB/s", flush=True)

if "dragon" in sys.argv[1:] or len(sys.argv) == 1:
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(ROOT, "tests", "golden", "final_avec_bonus.scene"))
    cfg.set_size(3840, 2160)
    rt.upload_scene(cfg.scene())
    run("dragon 4K", cfg.frame())
if "mesh" in sys.argv[1:] or len(sys.argv) == 1:
    syn = importlib.import_module("raytracer_rust_amd.synthetic")
    t0 = time.time(); scene = syn.product_scene(rt); t1 = time.time()
    info = rt.upload_scene(scene.c)
    print(f"mesh generated in {t1-t0:.1f} s, built in {info['build_ms']/1e3:.2f} s", flush=True)
    run("mesh 8K", syn.product_frame(rt, 7680, 4320))
