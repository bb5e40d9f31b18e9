#!/bin/bash
# Per-kernel durations of one frame of the headline workload (rocprofv3 kernel trace).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-t}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace_$TAG -o lv -- python3 $ROOT/scripts/render_once.py 3 > /dev/null 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$ROOT/gpurun_out/trace_$TAG/lv_kernel_trace.csv")))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("closest", "shade", "fold", "render_kernel"))]
last = rows[-11:] if len(rows) >= 11 else rows
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print(f'{name:28s} start {(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  dur {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  vgpr {r.get("VGPR_Count", "?")} lds {r.get("LDS_Block_Size", "?")} grid {r.get("Grid_Size", "?")}')
PY
