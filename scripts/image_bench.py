#!/usr/bin/env python3
"""Times the frame post-processing kernels (compare, RGB packing) on 8K frames and the host PNG writer.
Run under rocprofv3 --kernel-trace --stats for the kernel durations; the host clock here includes
launch and synchronisation."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
rt = ge.load_package()
import torch
rt.init(0)
W, H = 7680, 4320
a = torch.randint(-2**31, 2**31 - 1, (H, W), dtype=torch.int32, device="cuda:0")
b = a.clone(); b[::7, ::5] += 3
d = torch.zeros_like(a)
rgb = torch.zeros(3 * W * H, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
for name, fn, nbytes in (("compare+diff", lambda: rt.compare_frames(a.data_ptr(), (W, H), b.data_ptr(), (W, H), d.data_ptr()), 12 * W * H),
                         ("compare", lambda: rt.compare_frames(a.data_ptr(), (W, H), b.data_ptr(), (W, H)), 8 * W * H),
                         ("pack_rgb", lambda: (rt.pack_rgb(a.data_ptr(), W, H, rgb.data_ptr()), rt.synchronize()), 7 * W * H)):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {dt*1e6:.1f} us per call (host clock), {nbytes/dt/1e12:.2f} TB/s incl. launch+sync", flush=True)
host = np.empty((H, W), dtype=np.uint32)
t0 = time.perf_counter(); px = a.cpu().numpy().view(np.uint32); t1 = time.perf_counter()
print(f"D2H 4 B/pixel: {(t1-t0)*1e3:.1f} ms")
t0 = time.perf_counter(); r = rgb.cpu().numpy(); t1 = time.perf_counter()
print(f"D2H 3 B/pixel: {(t1-t0)*1e3:.1f} ms")
