#!/bin/bash
# Round profile on the GPU box: kernel statistics of the bench command (rocprofv3 --kernel-trace --stats) and the counter
# passes, for the dragon and for the mesh.  Results under gpurun_out/prof_<tag>/ ; copy what is to be kept into profiles/.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dragon_stats" -o s -- python3 "$ROOT/bench.py" --steps 60 --warmup 10 --no-cpu-baseline > "$OUT/dragon_bench.json" 2> "$OUT/dragon_bench.err" || { echo "dragon stats failed"; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mesh_stats" -o s -- python3 "$ROOT/bench.py" --config mesh --no-cpu-baseline > "$OUT/mesh_bench.json" 2> "$OUT/mesh_bench.err" || { echo "mesh stats failed"; exit 1; }
cd "$ROOT"
PASSES="fetch write sq1 sq2 tcp ta" bash scripts/profile_pmc.sh ${TAG}_dragon || exit 1
PASSES="fetch write sq1 sq2 tcp" bash scripts/profile_pmc.sh ${TAG}_mesh mesh || exit 1
python3 scripts/summarize_pmc.py gpurun_out/pmc_${TAG}_dragon "$OUT/dragon_pmc.json" > /dev/null
python3 scripts/summarize_pmc.py gpurun_out/pmc_${TAG}_mesh "$OUT/mesh_pmc.json" > /dev/null
echo done
