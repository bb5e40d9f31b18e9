// bvh_gpu.h -- interface of the GPU tree builder (bvh_gpu.hip).
#ifndef RT_BVH_GPU_H
#define RT_BVH_GPU_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "layout.h"

namespace rtg {

// Everything a scene blob is made of, in host memory; the tree shapes in any order.
struct SceneParts {
    const rtl::ShapeRec* tree; uint32_t n_tree;
    const rtl::ShapeRec* list; uint32_t n_list;
    const rtl::MaterialRec* mats; uint32_t n_mats;
    const rtl::LightRec* lights; uint32_t n_lights;
    uint32_t finite_shading;
    uint32_t tree_kinds;       // bit k set: `tree` holds shapes of kind k (goes into the blob header)
    float pad_abs;             // absolute outward padding of every node box (the host builder's rule)
    uint32_t depth_limit;      // deepest tree the render kernels can traverse
};

struct BuiltScene {
    void* d_blob; size_t bytes;       // hipMalloc'ed; the caller owns it
    rtl::BlobHeader header;
};

enum BuildStatus { BUILD_OK = 0, BUILD_FAILED = 1, BUILD_TOO_DEEP = 2 };

// Builds the tree on the GPU of the current device and assembles the scene blob in device memory.
// BUILD_TOO_DEEP: the tree came out deeper than depth_limit (the caller falls back to the host builder).
BuildStatus build_scene_on_gpu(const SceneParts& in, hipStream_t stream, BuiltScene* out, std::string* err);

}  // namespace rtg
#endif
