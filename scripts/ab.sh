#!/bin/bash
# A/B timing of side-by-side library builds on one GPU box: scripts/ab.sh <workload args of mesh_bench.py> -- <variant names...>
# ("cur" = the regular build).  Every variant runs in a process of its own, under its own time limit.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
for v in "$@"; do
    if [ "$v" = cur ]; then lib=""; else lib=$ROOT/raytracer-rust_amd/variants/librt_amd_$v.so; fi
    echo "== $v"
    RT_AMD_LIBRARY=$lib timeout -k 10 300 python3 "$ROOT/scripts/mesh_bench.py" "${args[@]}" || exit 1
done
