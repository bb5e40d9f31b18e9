#!/bin/bash
# Instruction counters of side-by-side library builds (scripts/build_variant.sh) over three 4K dragon frames:
#   scripts/variant_counters.sh cur name1 name2 ...      ("cur" = the regular build)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
    if [ "$v" = cur ]; then export RT_AMD_LIBRARY=""; else export RT_AMD_LIBRARY=$ROOT/raytracer-rust_amd/variants/librt_amd_$v.so; fi
    out=$ROOT/gpurun_out/vc_$v
    rm -rf "$out"
    timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out" -o p -- python3 "$ROOT/scripts/render_once.py" 3 > "$out.log" 2>&1 || { echo "$v: rocprofv3 failed"; exit 1; }
    python3 - "$v" "$out" <<'PY'
import csv, glob, sys, collections
name, out = sys.argv[1], sys.argv[2]
tot, n = collections.Counter(), collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel<false" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(name, {k: round(tot[k] / n[k] / 1e6, 2) for k in sorted(tot)}, "(millions per launch)")
PY
done
