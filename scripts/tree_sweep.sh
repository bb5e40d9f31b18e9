# Frame time of the 4K dragon against the host tree builder's cost model (RT_BVH_COST_SHAPE, RT_BVH_LEAF_CAP: diagnostic overrides).
for cs in ${COSTS:-0.5 0.75 1.0}; do for cap in ${CAPS:-8}; do
  echo "== COST_SHAPE=$cs LEAF_CAP=$cap"
  RT_BVH_COST_SHAPE=$cs RT_BVH_LEAF_CAP=$cap python3 scripts/ab_frames.py dragon --rounds 1 --frames 40 cur | tail -1
done; done
