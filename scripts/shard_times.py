#!/usr/bin/env python3
"""Kernel time of the 4K dragon frame cut into N shards, one launch per shard (diagnostic: what a rank of an
N-GPU job spends in its kernel, and how much a launch loses to ramp-up and tail)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
rt = ge.load_package()
import torch
cfg = rt.ParsedConfigState().load_config_file(os.path.join(ROOT, "tests", "golden", "final_avec_bonus.scene"))
cfg.set_size(3840, 2160)
rt.RayTracer(cfg)
f = cfg.frame()
buf = torch.zeros((2160, 3840), dtype=torch.int32, device="cuda:0")
for n in (1, 2, 4, 8):
    ts = []
    for s in range(n):
        best = min(rt.render_tiles(f, s, n, buf.data_ptr(), want_stats=True)["kernel_ms"] for _ in range(5))
        ts.append(best)
    print(f"{n} shards: " + " ".join(f"{t:.3f}" for t in ts) + f"  sum {sum(ts):.3f} ms  max {max(ts):.3f} ms  -> ideal {n}-GPU speed-up {ts[0] if n == 1 else 0:.0f}", flush=True)
