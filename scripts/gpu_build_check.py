#!/usr/bin/env python3
"""Builds the same scenes with the host and the GPU tree builder: frames must be identical, the GPU-built
blob must pass the structural check; prints build and render times (diagnostic)."""
import os, sys, time, zlib, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
rt = ge.load_package()
rt.init(0)

def run(name, scene, frame):
    out = {}
    for gpu in (False, True):
        t0 = time.perf_counter(); info = rt.upload_scene(scene, gpu_build=gpu); t1 = time.perf_counter()
        rt.validate_blob(rt.download_scene_blob())      # before any ray follows it
        img, st = rt.render(frame, count_work=True)
        ts = [rt.render(frame)[1]["kernel_ms"] for _ in range(4)]
        out[gpu] = img
        print(f"{name} {'gpu ' if gpu else 'host'} build: upload {1e3*(t1-t0):8.1f} ms  pairs {info['node_pairs']} leaves {info['leaves']} depth {info['max_depth']} "
              f"max leaf {info['max_leaf']} sah {info['sah_cost']:.1f}  kernel {min(ts):.3f} ms  boxes {st['nodes_visited']/1e6:.0f}M shapes {st['prims_tested']/1e6:.0f}M "
              f"crc {zlib.crc32(img.tobytes()):08x}", flush=True)
    assert np.array_equal(out[False], out[True]), "frames differ between builders"

cfg = rt.ParsedConfigState().load_config_file(os.path.join(ROOT, "tests", "golden", "final_avec_bonus.scene"))
run("dragon 640x480", cfg.scene(), cfg.frame())
cfg.set_size(3840, 2160)
run("dragon 4K", cfg.scene(), cfg.frame())
cfg2 = rt.ParsedConfigState().load_config_file(os.path.join(ROOT, "tests", "golden", "test.scene"))
run("test.scene", cfg2.scene(), cfg2.frame())
if "mesh" in sys.argv[1:]:
    syn = importlib.import_module("raytracer_rust_amd.synthetic")
    scene = syn.product_scene(rt)
    run("mesh 8K", scene.c, syn.product_frame(rt, 7680, 4320))
