#!/bin/bash
# Kernel time of the 4K dragon frame for a few builder settings (diagnostic).
for cfg in "1 1 8" "1 1 4" "1 1 2" "1 2 8" "2 1 8" "1 3 8" "1 1.5 8" "0.5 1 8"; do
  set -- $cfg
  echo "traverse=$1 shape=$2 maxleaf=$3: $(RT_AMD_BVH_COST_TRAVERSE=$1 RT_AMD_BVH_COST_SHAPE=$2 RT_AMD_BVH_MAX_LEAF=$3 timeout -k 10 100 python3 scripts/render_once.py 5 2>/dev/null | tail -1)"
done
