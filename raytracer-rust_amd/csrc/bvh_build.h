// bvh_build.h -- host builder of the GPU bounding volume hierarchy (see bvh_build.cpp).
#ifndef RT_BVH_BUILD_H
#define RT_BVH_BUILD_H

#include <stdint.h>
#include <string>
#include <vector>

#include "layout.h"
#include "no_init_vector.h"

namespace rtb {

struct BvhResult {
    rtu::NoInitVector<rtl::NodePair> pairs;   // 64-byte child-pair records; pairs[root_ref] is the root when root_ref is inner
    rtu::NoInitVector<uint32_t> order;        // tree-shape slot -> index of the input box (leaves are contiguous slots)
    uint32_t root_ref;                  // REF_EMPTY for no shapes, a leaf reference for a single-leaf tree
    uint32_t max_depth;                 // node pairs on the longest root-to-leaf path = traversal stack bound
    uint32_t max_leaf;
    uint32_t n_leaves;
    double sah_cost;
};

// lo / hi: n x 3 axis-aligned boxes of the shapes (shape.rs:205-234 for what a box is).
// depth_limit: longest root-to-leaf path allowed (the traversal stack of the kernels).
// pad_abs: absolute outward padding added to every stored box besides the relative one.
bool build_bvh(const float* lo, const float* hi, uint32_t n, uint32_t depth_limit, float pad_abs,
               BvhResult* out, std::string* err);

}  // namespace rtb
#endif
