#!/usr/bin/env python3
"""Turns a summary of rocprofv3 --pmc passes (scripts/summarize_pmc.py) into the entry of profiles/traffic.json that
bench.py reads for one workload:  python scripts/make_traffic.py <dragon|mesh> <summary.json> "<workload text>" """
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, summary, workload = sys.argv[1], json.load(open(sys.argv[2])), sys.argv[3]
path = os.path.join(ROOT, "profiles", "traffic.json")
try:
    table = json.load(open(path))
except (OSError, ValueError):
    table = {}
import subprocess
try:
    commit = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h"], capture_output=True, text=True, timeout=10).stdout.strip()
except (OSError, subprocess.SubprocessError):
    commit = ""
entry = {"workload": workload, "kernel": "render_kernel", "profiled_at_commit": commit or os.environ.get("RT_PROFILED_COMMIT", "unknown"), "source": os.path.relpath(sys.argv[2], ROOT) +
         ": rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum and the SQ counters in separate passes over "
         "scripts/render_once.py, means per launch; FETCH_SIZE (KiB) doubled as the gfx950 correction for 128-byte requests tallied "
         "at 64 bytes, WRITE_SIZE (KiB) taken as is"}
if "hbm_read_bytes_x2" in summary:
    entry["read_bytes_fetch_size_x2"] = summary["hbm_read_bytes_x2"]
    entry["read_bytes_fetch_size_raw"] = summary["hbm_read_bytes_raw"]
    entry["write_bytes"] = summary.get("hbm_write_bytes")
    entry["hbm_bytes_per_launch"] = summary["hbm_read_bytes_x2"] + (summary.get("hbm_write_bytes") or 0.0)
for src, dst in (("l2_hit_rate", "l2_hit_rate"), ("valu_lane_utilisation", "valu_lane_utilisation"), ("SQ_INSTS_VALU", "valu_insts"),
                 ("SQ_INSTS_SALU", "salu_insts"), ("SQ_INSTS_VMEM_RD", "vmem_read_insts"), ("SQ_WAVE_CYCLES", "wave_quad_cycles"),
                 ("SQ_WAIT_ANY", "wait_any_quad_cycles"), ("GRBM_GUI_ACTIVE", "gui_active_cycles_all_xcds")):
    if src in summary:
        entry[dst] = summary[src]
table[name] = entry
json.dump(table, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
