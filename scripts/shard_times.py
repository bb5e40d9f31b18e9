#!/usr/bin/env python3
"""Kernel time of the 4K dragon frame cut into N shards, one launch per shard (diagnostic: what a rank of an
N-GPU job spends in its kernel), serial on one lane and pipelined over two and four lanes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
rt = ge.load_package()
import torch
cfg = rt.ParsedConfigState().load_config_file(os.path.join(ROOT, "tests", "golden", "final_avec_bonus.scene"))
cfg.set_size(3840, 2160)
rt.RayTracer(cfg)
f = cfg.frame()
bufs = [torch.zeros((2160, 3840), dtype=torch.int32, device="cuda:0") for _ in range(rt.LANES)]
for n in (1, 2, 4, 8):
    ts, piped = [], []
    for s in range(n):
        rt.select_lane(0)
        ts.append(min(rt.render_tiles(f, s, n, bufs[0].data_ptr(), want_stats=True)["kernel_ms"] for _ in range(5)))
    s = 0
    for lanes in (2, 4):
        for reps in (12, 80):           # warm the lanes' region orders, then time
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                rt.select_lane(i % lanes)
                rt.render_tiles(f, s, n, bufs[i % lanes].data_ptr())
            torch.cuda.synchronize()
            piped.append((time.perf_counter() - t0) / reps * 1e3)
    rt.select_lane(0)
    print(f"{n} shards: serial " + " ".join(f"{t:.3f}" for t in ts) + f"  (sum {sum(ts):.3f} ms);  shard 0 per frame, pipelined over two lanes: {piped[1]:.3f} ms, "
          f"over four: {piped[3]:.3f} ms", flush=True)
