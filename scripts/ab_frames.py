#!/usr/bin/env python3
"""A/B timing of side-by-side builds of the library (scripts/build_variant.sh) on one GPU box, less noisy than scripts/ab.sh:
every variant renders the workload in a process of its own, several rounds interleaved, and the kernel times of all of a
variant's frames (after warm-up) are pooled: min, median, mean.
  python3 scripts/ab_frames.py [dragon|mesh] [--rounds 3] [--frames 40] cur name1 name2 ...      ("cur" = the regular build)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, importlib, json
sys.path.insert(0, %r)
import __graft_entry__ as ge
rt = ge.load_package(); rt.init(0)
what, frames = sys.argv[1], int(sys.argv[2])
if what == "mesh":
    syn = importlib.import_module("raytracer_rust_amd.synthetic")
    scene = syn.product_scene(rt); rt.upload_scene(scene.c, gpu_build=True); f = syn.product_frame(rt, 7680, 4320)
else:
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(%r, "tests", "golden", "final_avec_bonus.scene"))
    cfg.set_size(3840, 2160); rt.upload_scene(cfg.scene()); f = cfg.frame()
for _ in range(6): rt.render_frame(f, want_stats=True)
print(json.dumps([rt.render_frame(f, want_stats=True)[1]["kernel_ms"] for _ in range(frames)]))
''' % (ROOT, ROOT)
args = sys.argv[1:]
what = "dragon"
if args and args[0] in ("dragon", "mesh"): what = args.pop(0)
rounds, frames = 3, 40
while args and args[0].startswith("--"):
    k = args.pop(0); v = int(args.pop(0))
    if k == "--rounds": rounds = v
    elif k == "--frames": frames = v
times = {v: [] for v in args}
for r in range(rounds):
    for v in args:
        env = dict(os.environ)
        env["RT_AMD_LIBRARY"] = "" if v == "cur" else os.path.join(ROOT, "raytracer-rust_amd", "variants", f"librt_amd_{v}.so")
        out = subprocess.run([sys.executable, "-c", CHILD, what, str(frames)], env=env, capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            print(v, "failed:", out.stderr[-2000:], flush=True); sys.exit(1)
        t = json.loads(out.stdout.strip().splitlines()[-1]); times[v] += t
        print(f"round {r} {v:24s} min {min(t):.3f} median {sorted(t)[len(t)//2]:.3f}", flush=True)
print()
for v, t in times.items():
    t = sorted(t)
    print(f"{v:24s} min {t[0]:.3f}  median {t[len(t)//2]:.3f}  mean {sum(t)/len(t):.3f}  ({len(t)} frames)")
