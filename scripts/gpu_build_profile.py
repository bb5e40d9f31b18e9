#!/usr/bin/env python3
"""Builds the procedural ten-million-triangle mesh on the GPU twice (diagnostic, to be run under rocprofv3)."""
import os, sys, time, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
rt = ge.load_package()
rt.init(0)
syn = importlib.import_module("raytracer_rust_amd.synthetic")
scene = syn.product_scene(rt)
for _ in range(2):
    t0 = time.perf_counter(); info = rt.upload_scene(scene.c, gpu_build=True); t1 = time.perf_counter()
    print(f"upload {1e3 * (t1 - t0):.1f} ms", info, flush=True)
