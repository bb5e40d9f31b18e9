#!/bin/bash
# Same-box A/B of library builds on the 4K dragon: scripts/ab_cold.sh <variant names...>  ("cur" = the regular build).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for v in "$@"; do
    if [ "$v" = cur ]; then lib=""; else lib=$ROOT/raytracer-rust_amd/variants/librt_amd_$v.so; fi
    echo "== $v"
    RT_AMD_LIBRARY=$lib timeout -k 10 120 python3 "$ROOT/scripts/cold_frames.py" || exit 1
done
