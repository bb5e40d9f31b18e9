#!/bin/bash
# scripts/ab_occ.sh <variants...>: every variant (built with -DRT_DIAG_LDS_PAD=1) at full occupancy and at one wavefront per SIMD
# (LDS padded to 40 KiB per wavefront), where the kernel time is the latency chain of a single wavefront.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for v in "$@"; do
    for pad in 0 40900; do
        echo "== $v pad $pad"
        RT_DIAG_LDS_PAD=$pad RT_AMD_LIBRARY=$ROOT/raytracer-rust_amd/variants/librt_amd_$v.so timeout -k 10 300 python3 "$ROOT/scripts/mesh_bench.py" dragon || exit 1
    done
done
