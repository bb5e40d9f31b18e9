#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc passes written by scripts/profile_pmc.sh: per counter, the mean
over the dispatches of the render kernel.  Usage: python scripts/summarize_pmc.py gpurun_out/pmc_<tag> [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
KERNELS = ("render_kernel", "closest_kernel", "shade_kernel", "fold_kernel")
acc = defaultdict(list)          # counter -> one sum per pass file that collected it
per_kernel = defaultdict(lambda: defaultdict(float))      # kernel -> counter -> sum over the dispatches of ONE frame (the last)
for path in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    per_dispatch = defaultdict(float)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            kn = next((k for k in KERNELS if k in row["Kernel_Name"]), None)
            if kn is None:
                continue
            per_dispatch[(int(row["Dispatch_Id"]), row["Counter_Name"], kn)] += float(row["Counter_Value"])
    # dispatches of the last frame: everything after the second-to-last fold_kernel (or the last render_kernel)
    ids = sorted({d for d, _, _ in per_dispatch})
    folds = sorted({d for d, _, k in per_dispatch if k in ("fold_kernel", "render_kernel")})
    lo = folds[-2] if len(folds) >= 2 else -1
    in_file = defaultdict(float)
    for (d, name, kn), v in per_dispatch.items():
        if d > lo:
            in_file[name] += v
            per_kernel[kn][name] = v if name not in per_kernel[kn] else per_kernel[kn][name]      # (a counter collected in two passes is kept once)
    for name, v in in_file.items():
        acc[name].append(v)
out = {k: sum(v) / len(v) for k, v in acc.items()}
d = dict(out)
if "FETCH_SIZE" in d:
    # FETCH_SIZE/WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte requests of wide
    # coalesced reads at 64 bytes, so it is doubled for such streams; the render kernel's reads are
    # 16-byte-per-lane gathers, for which the factor is uncalibrated: both figures are kept.
    d["hbm_read_bytes_raw"] = d["FETCH_SIZE"] * 1024
    d["hbm_read_bytes_x2"] = d["FETCH_SIZE"] * 2048
if "WRITE_SIZE" in d:
    d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
    d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
if "SQ_THREAD_CYCLES_VALU" in d and "SQ_ACTIVE_INST_VALU" in d:
    d["valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_ACTIVE_INST_VALU"] * 64)
if "SQ_ACTIVE_INST_VALU" in d and "GRBM_GUI_ACTIVE" in d:
    pass
d["per_kernel"] = {k: dict(v) for k, v in per_kernel.items()}
for k, v in d["per_kernel"].items():
    if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
        v["valu_lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)
print(json.dumps(d, indent=1))
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        json.dump(d, fh, indent=1)
