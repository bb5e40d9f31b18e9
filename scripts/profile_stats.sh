#!/bin/bash
# The kernel-statistics half of scripts/profile_all.sh (rocprofv3 --kernel-trace --stats over the bench command, dragon and
# mesh) plus the two plain bench lines; for changes that leave the render kernel, and so the counter passes, as they were.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dragon_stats" -o s -- python3 "$ROOT/bench.py" --steps 60 --warmup 10 --no-cpu-baseline > "$OUT/dragon_bench.json" 2> "$OUT/dragon_bench.err" || { echo "dragon stats failed"; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mesh_stats" -o s -- python3 "$ROOT/bench.py" --config mesh --no-cpu-baseline > "$OUT/mesh_bench.json" 2> "$OUT/mesh_bench.err" || { echo "mesh stats failed"; exit 1; }
cd "$ROOT"
timeout -k 10 400 python3 bench.py > "$OUT/dragon_bench_plain.json" 2> "$OUT/dragon_bench_plain.err" || { echo "dragon bench failed"; exit 1; }
timeout -k 10 600 python3 bench.py --config mesh > "$OUT/mesh_bench_plain.json" 2> "$OUT/mesh_bench_plain.err" || { echo "mesh bench failed"; exit 1; }
echo done
